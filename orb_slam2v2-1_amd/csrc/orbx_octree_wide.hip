// orbx_octree_wide.hip - the quad-tree kernels of orbx_octree.hip built a second time with 1024-thread workgroups
// (k_octree_pyr_wide, k_octree_big_wide, k_octree_wide).  One workgroup owns one pyramid level of one image and its two key sweeps are
// bound by the LDS of its CU: for the large levels of large images (1920x1080 level 0: ~70 k keys) twice the threads shorten that
// critical path (174 -> 150 us); for 1241x376-sized levels 512 threads pack better at batch 128 (orbx_extract_dev.h).
#define OCT_T 1024
#define k_octree_pyr k_octree_pyr_wide
#define k_octree_big k_octree_big_wide
#define k_octree k_octree_wide
#include "orbx_octree.hip"
