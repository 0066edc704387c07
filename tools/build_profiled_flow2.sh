# builds libvdo_hip.so with the phase profiler of k_flow2_lm compiled in (debug aid; rebuild normally afterwards: touch flow2.hip && make)
cd "$(dirname "$0")/../vdo_slam_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-value -DF2_PROFILE $F2_EXTRA -c flow2.hip -o flow2.o && make 2>&1 | tail -1
