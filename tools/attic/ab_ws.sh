# Same-box A/B of the two round-5 GEMM experiments: fused epilogues (ops.FUSE_EPILOGUES) x weight-stationary kernel (a library built
# with -DDGDM_WS: bash tools/build_variant_lib.sh ws -DDGDM_WS).  gpurun -- bash tools/ab_ws.sh
B="--steps 40 --no-cpu-baseline --no-strict --no-raster --no-gather --sustain-seconds 0"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], r["value"], r["ms_per_step"])'
for rep in 1 2; do
python tools/bench_with.py FUSE_EPILOGUES=True -- $B 2>/dev/null | python -c "$P" "nows+fused"
python tools/bench_with.py FUSE_EPILOGUES=False -- $B 2>/dev/null | python -c "$P" "nows+unfused"
python tools/run_with_lib.py dgdm_histopath_lab_amd/lib/ws/libdgdm_hip.so tools/bench_with.py FUSE_EPILOGUES=True -- $B 2>/dev/null | python -c "$P" "ws+fused"
python tools/run_with_lib.py dgdm_histopath_lab_amd/lib/ws/libdgdm_hip.so tools/bench_with.py FUSE_EPILOGUES=False -- $B 2>/dev/null | python -c "$P" "ws+unfused"
done
