mkdir -p gpurun_out/r05q
python -m pytest tests/test_hip_switches.py -q -s > gpurun_out/r05q/switches.txt 2>&1; tail -25 gpurun_out/r05q/switches.txt | cut -c1-160
