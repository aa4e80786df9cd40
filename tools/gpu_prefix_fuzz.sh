# Round 6: tools/prefix_fuzz.py -- first against a build WITHOUT seg_resume's first-unit fix (the fuzz must find it), then 20 seeds.
cd $GRAFT_REPO_ROOT
echo "=== without the fix (BRX_NO_FIRST_UNIT_FIX): seeds 1 .. 4"
BRX_DEFS="BRX_NO_FIRST_UNIT_FIX" python brotli-rs_amd/build.py --force > /dev/null 2>&1
for s in 1 2 3 4; do timeout 280 python tools/prefix_fuzz.py 30 $s 2>&1 | grep -E "MISMATCH|prefix_fuzz seed" | tail -4; done
echo "=== with the fix: seeds 1 .. 20, 30 rounds each"
python brotli-rs_amd/build.py --force > /dev/null 2>&1
for s in $(seq 1 20); do timeout 280 python tools/prefix_fuzz.py 30 $s 2>&1 | grep -E "MISMATCH|prefix_fuzz seed|Error|error" | tail -6; done
