cd $GRAFT_REPO_ROOT
for env in "X=0" "DFHIP_JOBMUL=2" "DFHIP_JOBMUL=4" "DFHIP_MAXCHUNKS=4" "DFHIP_MAXCHUNKS=8" "DFHIP_JOBMUL=2 DFHIP_MAXCHUNKS=4" "DFHIP_LDS_SINGLE=76"; do
  echo "== $env"; env $env python tools/conv_bench.py --wide --only w_l1,w_l2a,w_l2,w_l3a,w_l3,w_l4a,w_l4 --iters 20 2>&1 | grep -E " fwd  | fwd_stats" 
done
