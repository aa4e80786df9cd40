# Round 6: the ring reload's first-unit fix (seg_resume, brx_kernels.hip) A/B -- the regression test without the fix (must fail), then
# with it, the fuzz seeds that found it, and the whole GPU suite.
cd $GRAFT_REPO_ROOT
T=tests/test_gpu_parity.py::test_ring_reload_in_the_first_two_kib_keeps_the_streams_first_bytes
echo "=== without the fix (BRX_NO_FIRST_UNIT_FIX)"
BRX_DEFS="BRX_NO_FIRST_UNIT_FIX" python brotli-rs_amd/build.py --force > /dev/null 2>&1
timeout 300 python -m pytest $T -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|assert not bad|^E  " | head -12
echo "=== with the fix"
python brotli-rs_amd/build.py --force > /dev/null 2>&1
timeout 300 python -m pytest $T -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E  " | head -12
for seed in 6 7 43 9 10 11 12; do
  timeout 600 python tools/node_fuzz.py 30 $seed 2>&1 | grep -E "MISMATCH|node_fuzz seed"
done
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 ) > gpurun_out/r06_suite_final.log 2>&1
grep -E "passed|failed" gpurun_out/r06_suite_final.log; grep real gpurun_out/r06_suite_final.log
