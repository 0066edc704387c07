cd $GRAFT_REPO_ROOT/vdo_slam_amd/csrc
for w in 4 5 6; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -DVDO_SWEEP_WAVES=$w -c ba_sweep.hip -o ba_sweep.o 2>&1 | grep -E "error" 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvdo_hip.so *.o -ldl
  for ss in 64 40 28; do echo "waves $w soft_slots $ss"; (cd ../..; VDO_BA_SOFT_SLOTS=$ss python tools/sweep_only.py 2>&1 | grep "^n_pose" | sed 's/n_pose 2190 dims//'); done
done
