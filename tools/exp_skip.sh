# timing experiments on the diagnostic build: FRZ_WF_SKIP bit i leaves store group i out (results are wrong, timing only)
mkdir -p gpurun_out
for S in ${SKIPS:-0 1 2 4 8 16 32 63}; do
  echo "skip=$S" >> gpurun_out/exp.log
  FRZ_HIP_LIB=$PWD/free-range-zoo_amd/csrc/libfrz_hip_stamps.so FRZ_WF_SKIP=$S timeout -k 10 100 python tools/kfloor.py 65536 2>/dev/null | tail -1 >> gpurun_out/exp.log || exit 1
done
cat gpurun_out/exp.log
