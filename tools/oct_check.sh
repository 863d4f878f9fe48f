# quad-tree change check on the GPU box: parity tests of every quad-tree form, the per-pass time stamps at 1920x1080, then the step at three sizes
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_extract_gpu.py -m gpu -x -q -k "staged or forms or quadtree or octree or small_batch or product" 2>&1 | tail -2 || exit 1
ORBX_LIB=$GRAFT_REPO_ROOT/orb_slam2v2-1_amd/lib/liborbx_hip_dev.so python3 tools/octree_pass_time_probe.py fullhd4000 2>&1 | grep -v amdgpu | tail -8
LIBS="-" WL="mono_1920x1080_4000feat:64 mono_1920x1080_4000feat:32 kitti_stereo_1241x376_1000feat:64 kitti_stereo_1241x376_2000feat:64" VERIFY=" " bash tools/ab_libs.sh
