cd $GRAFT_REPO_ROOT
BRX_BRINGUP=1 python brotli-rs_amd/build.py --force > /dev/null 2>&1
g++ -O2 -std=c++17 tools/diag_main.cpp -o tools/diag_main -Lbrotli-rs_amd -lbrx -Wl,-rpath,$GRAFT_REPO_ROOT/brotli-rs_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
G=tests/golden/data
for f in monkey asyoulik.txt; do echo "== $f"; BRX_DEBUG_STATS=1 timeout 120 ./tools/diag_main $G/$f.compressed 1048592 1 2 2>&1 | grep -v amdgpu | tail -8; done
