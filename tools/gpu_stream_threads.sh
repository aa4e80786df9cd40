# The Read facade under threads (tests/cpp/stream_threads.cpp): rate and batch sizes for 1 .. 512 host threads.
cd $GRAFT_REPO_ROOT
L=brotli-rs_amd
g++ -O2 -std=c++17 tests/cpp/stream_threads.cpp -o /tmp/stream_threads -L $L -lbrx -Wl,-rpath,$PWD/$L -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -lpthread || exit 1
D=tests/golden/data
for T in 1 16 64 256 512; do timeout 300 /tmp/stream_threads $D/alice29.txt.compressed $D/alice29.txt $T $((T < 16 ? 50 : 20)) 2>&1 | grep -v amdgpu.ids; done
for T in 64 512; do timeout 300 /tmp/stream_threads $D/monkey.compressed $D/monkey $T 50 2>&1 | grep -v amdgpu.ids; done
