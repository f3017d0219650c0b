#!/bin/bash
# round 3, call H: clean A/B of the lane-generated twiddles (two builds), whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/r3h_tw.txt
for v in table shuffle table shuffle; do
  if [ $v = shuffle ]; then export CHZ_LIB=$GRAFT_REPO_ROOT/ka9q-radio_amd/libchz_hip_twshuffle.so; else unset CHZ_LIB; fi
  timeout 120 python scripts/kernel_times.py "twiddles=$v" 2>/dev/null | tail -1 >> gpurun_out/r3h_tw.txt
done
unset CHZ_LIB
cut -c1-260 gpurun_out/r3h_tw.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r3h_pytest.log
