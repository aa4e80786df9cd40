# Facade locking stress, harder than the test: tests/cpp/stream_mix.cpp with 64 / 128 / 16 threads, several runs.
cd $GRAFT_REPO_ROOT
L=brotli-rs_amd; D=tests/golden/data
g++ -O2 -std=c++17 tests/cpp/stream_mix.cpp -o /tmp/stream_mix -L $L -lbrx -Wl,-rpath,$PWD/$L -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -lpthread || exit 1
head -c 30000 $D/alice29.txt.compressed > /tmp/cut.compressed
A=""
for n in alice29.txt monkey quickfox_repeated compressed_repeated empty x asyoulik.txt lcet10.txt; do A="$A $D/$n.compressed $D/$n"; done
for run in "64 100" "128 40" "16 200" "64 100" "256 20"; do timeout 300 /tmp/stream_mix $run $A /tmp/cut.compressed -24 2>&1 | grep -v amdgpu.ids; echo "rc=$?"; done
