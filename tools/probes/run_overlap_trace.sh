mkdir -p gpurun_out/r05d
python -m pytest tests/test_hip_lazy_records.py -q > gpurun_out/r05d/lazy_tests.txt 2>&1
tail -25 gpurun_out/r05d/lazy_tests.txt
cd /tmp && export TMPDIR=/tmp
export STEPS=8 DFHIP_OVERLAP_CUS=128 DFHIP_WGRAD_SIDE=4
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05d/trace -- python3 $GRAFT_REPO_ROOT/tools/probes/overlap_eager.py > $GRAFT_REPO_ROOT/gpurun_out/r05d/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r05d/trace -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find the last k_prof_gate kernel, print what follows
idx=[i for i,r in enumerate(rows) if 'k_prof_gate' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['End_Timestamp'])
out=open('gpurun_out/r05d/last_step_timeline.txt','w')
for r in rows[i0:]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    out.write('%9.1f %9.1f %7.1f q%-3s %s\n'%(s,e,e-s,r.get('Queue_Id','?'),r['Kernel_Name'][:90]))
out.close()
PY
rm -rf gpurun_out/r05d/trace/*/*.db 2>/dev/null
cat gpurun_out/r05d/last_step_timeline.txt | head -120
