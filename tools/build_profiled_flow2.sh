# builds vdo_slam_amd/libvdo_hip_prof.so = the product library with the phase profiler of k_flow2_lm compiled in (debug aid;
# load it with VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_prof.so python tools/flow2_phase_probe.py ...; the product library is not touched)
cd "$(dirname "$0")/../vdo_slam_amd/csrc" && make 2>&1 | tail -1 && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-value -DF2_PROFILE -DF2_THREADS=${F2_THREADS:-256} $F2_EXTRA -c flow2.hip -o /tmp/flow2_prof.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvdo_hip_prof.so $(ls *.o | grep -v '^flow2.o$') /tmp/flow2_prof.o -ldl && ls -la ../libvdo_hip_prof.so
