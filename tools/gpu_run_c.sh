cd $GRAFT_REPO_ROOT
g++ -O2 -std=c++17 tools/diag_main.cpp -o tools/diag_main -Lbrotli-rs_amd -lbrx -Wl,-rpath,$GRAFT_REPO_ROOT/brotli-rs_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 2>&1 | tail -3
tools/gpu_pcsamp2.sh tests/golden/data/backward65536.compressed 65808 4096 300 bw
tools/gpu_pcsamp2.sh tests/golden/data/quickfox_repeated.compressed 176144 8192 100 qr
tools/gpu_pcsamp2.sh tests/golden/data/alice29.txt.compressed 152096 4096 20 al
cd $GRAFT_REPO_ROOT; python tools/gpu_small_lat.py 2>&1 | tail -12
