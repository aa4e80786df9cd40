cd $GRAFT_REPO_ROOT
for defs in "BRX_SPEC_CK_MIN_DWORDS=64u" "BRX_SPEC_CK_MIN_DWORDS=300u"; do
  echo "=== [$defs]"
  BRX_DEFS="$defs" python brotli-rs_amd/build.py --force > /dev/null 2>&1
  python tools/node_fuzz_repro.py 7 26 132 2>&1 | grep -v amdgpu | grep -A1 "slot offset" | cut -c1-200
done
