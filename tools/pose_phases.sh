# builds nothing: expects tools/ab/pose_timing.so (hipcc ... -DORBFE_POSE_TIMING), swaps it in for one run
cp orbslam2_amd/liborbfe.so /tmp/liborbfe.keep && cp tools/ab/pose_timing.so orbslam2_amd/liborbfe.so && timeout -k 10 120 python3 tools/pose_phases.py; rc=$?; cp /tmp/liborbfe.keep orbslam2_amd/liborbfe.so; exit $rc
