// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns the ORB kernels use.
// MI355X_MICROARCH.md ("HBM") calibrates the counters for 16-B-per-lane streaming only (FETCH_SIZE reads 1/2)
// and says every other width must be calibrated on a known byte count.  Each kernel below touches a known
// number of bytes of a buffer far larger than the 256 MiB Infinity Cache, once:
//   cal_read16   16 B/lane streaming read                       (the guide's reference case)
//   cal_read4    4 B/lane streaming read                        (k_pyr_level, k_gather)
//   cal_window   36-row x 11-dword windows on a 30-px grid of a 1280-byte-pitch image, two dwords per lane
//                (k_fast_cells' staging; unique bytes = the whole image, windows overlap by 6 px)
//   cal_tile<W>  32-row tiles of W-byte rows (W = 64, 32) at a 1280-byte pitch that partition the buffer: no overlap
//   cal_write16  16 B/lane streaming store
//   cal_write4s  one dword store per 64 bytes                   (k_fast_cells' candidate slots: sparse, narrow)
// Build: hipcc --offload-arch=gfx950 -O2 tools/pmc_calibrate.hip -o tools/_build/pmc_calibrate
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- tools/_build/pmc_calibrate   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void cal_read16(const uint4 *__restrict__ p, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void cal_read4(const uint32_t *__restrict__ p, size_t n4, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}

// one wave per 30x30 cell; window = 36 rows x 11 dwords starting at the cell's aligned column - the shape of
// k_fast_cells' staging loop (two dwords per lane, all issued before use)
__global__ __launch_bounds__(256) void cal_window(const uint8_t *__restrict__ img, int pitch, int rows, int cellsX, int cellsY,
                                                  size_t imgBytes, uint32_t *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + wave;
    if (c >= cellsX * cellsY) return;
    const int cy = c / cellsX, cx = c - cy * cellsX;
    const uint8_t *base = img + (size_t)blockIdx.y * imgBytes + (size_t)(cy * 30) * pitch + ((cx * 30) & ~3);
    uint32_t acc = 0;
    for (int i = lane; i < 36 * 10; i += 64) {
        const int r = i / 10, q = i - r * 10;
        if (cy * 30 + r < rows) {
            const uint32_t *p = (const uint32_t *)(base + (size_t)r * pitch) + q;
            acc += p[0] ^ p[1];
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

// tiles that PARTITION the buffer (no overlap, every byte read exactly once): rows of TW bytes at a 1280-byte pitch,
// one wave per 32-row tile - how are row segments narrower than 128 B tallied?
template <int TW>
__global__ __launch_bounds__(256) void cal_tile(const uint8_t *__restrict__ buf, int pitch, size_t ntiles, uint32_t *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t t = (size_t)blockIdx.x * 4 + wave;
    if (t >= ntiles) return;
    constexpr int DW = TW / 4, RPI = 64 / DW;          // dwords per row, rows per iteration
    const int tilesX = pitch / TW;
    const size_t ty = t / tilesX, tx = t - ty * tilesX;
    const uint8_t *base = buf + ty * 32 * (size_t)pitch + tx * TW;
    uint32_t acc = 0;
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += RPI) {
        const int r = r0 + lane / DW, q = lane % DW;
        acc += ((const uint32_t *)(base + (size_t)r * pitch))[q];
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void cal_write16(uint4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

__global__ __launch_bounds__(256) void cal_write4s(uint32_t *__restrict__ p, size_t n64) {   // one dword per 64-byte line
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n64; i += (size_t)gridDim.x * 256) p[i * 16] = (uint32_t)i;
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB: four times the Infinity Cache
    uint8_t *buf;
    uint32_t *sink;
    CK(hipMalloc(&buf, bytes + 4096));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes + 4096));
    CK(hipDeviceSynchronize());
    const int pitch = 1280, rows = 390, cellsX = (pitch - 40) / 30, cellsY = rows / 30;   // 1280x390 "level 0" images
    const size_t imgBytes = (size_t)pitch * rows;
    const int nimg = (int)(bytes / imgBytes);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(cal_read16, dim3(8192), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
        hipLaunchKernelGGL(cal_read4, dim3(8192), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, sink);
        hipLaunchKernelGGL(cal_window, dim3((cellsX * cellsY + 3) / 4, nimg), dim3(256), 0, 0, buf, pitch, rows, cellsX, cellsY,
                           imgBytes, sink);
        {
            const size_t nrows32 = bytes / ((size_t)pitch * 32);
            const size_t nt64 = nrows32 * (pitch / 64), nt32 = nrows32 * (pitch / 32);
            hipLaunchKernelGGL(cal_tile<64>, dim3((unsigned)((nt64 + 3) / 4)), dim3(256), 0, 0, buf, pitch, nt64, sink);
            hipLaunchKernelGGL(cal_tile<32>, dim3((unsigned)((nt32 + 3) / 4)), dim3(256), 0, 0, buf, pitch, nt32, sink);
        }
        hipLaunchKernelGGL(cal_write16, dim3(8192), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
        hipLaunchKernelGGL(cal_write4s, dim3(8192), dim3(256), 0, 0, (uint32_t *)buf, bytes / 64);
        CK(hipDeviceSynchronize());
    }
    // expected unique bytes per launch
    const double win_unique = (double)nimg * (double)(cellsY * 30 + 6 > rows ? rows : cellsY * 30 + 6) * (cellsX * 30 + 16);
    const size_t tile_bytes = bytes / ((size_t)pitch * 32) * (size_t)pitch * 32;
    printf("{\"cal_tile<64>\": %zu, \"cal_tile<32>\": %zu, ", tile_bytes, tile_bytes);
    printf("\"cal_read16\": %zu, \"cal_read4\": %zu, \"cal_window\": %.0f, \"cal_write16\": %zu, \"cal_write4s_lines64\": %zu, \"cal_write4s_dwords\": %zu}\n",
           bytes, bytes, win_unique, bytes, bytes, bytes / 16);
    (void)hipFree(buf);
    (void)hipFree(sink);
    return 0;
}
