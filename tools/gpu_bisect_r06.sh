cd $GRAFT_REPO_ROOT
for defs in "" "BRX_NO_PAR_CLCODE" "BRX_NO_SKIP_END" "BRX_NO_PERIOD_COPY"; do
  echo "=== defs: [$defs]"
  BRX_DEFS="$defs" python brotli-rs_amd/build.py --force > /dev/null 2>&1
  python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "all_data_fixtures or encoder_streams_batch or config5_streams or generator_round_trip" 2>&1 | grep -E "^E  |passed|failed" | head -8
done
