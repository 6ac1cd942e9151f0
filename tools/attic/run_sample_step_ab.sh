set -e
cd /root/repo; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_sample.py -x -q -m gpu > gpurun_out/r06i_sample_tests.log 2>&1
R=320,8192,10000,20000,40000
timeout -k 10 200 python tools/microbench_sample_step.py $R >> gpurun_out/r06i_sample_step_ab.txt 2>&1
for L in stag64 stag128 stag192; do timeout -k 10 200 python tools/run_with_lib.py dgdm_histopath_lab_amd/lib/$L/libdgdm_hip.so tools/microbench_sample_step.py $R >> gpurun_out/r06i_sample_step_ab.txt 2>&1; done
tail -3 gpurun_out/r06i_sample_tests.log; cat gpurun_out/r06i_sample_step_ab.txt
