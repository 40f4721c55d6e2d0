/* CPU oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the reference's
 * dynamic_voxelize (mmdet3d/ops/voxel/src/voxelization_cpu.cpp:6-40 kernel body,
 * :139-168 grid size; CUDA twin voxelization_cuda.cu:22-63).  This fork CLAMPS
 * out-of-range points into the border cell (cpu.cpp:22-31) instead of writing -1.
 * Build: make -C oracle  ->  oracle/libgeomae_oracle.so  (loaded via ctypes by oracle_c.py).
 */
#include <math.h>
#include <stdint.h>

void geomae_oracle_grid_size(const float *voxel_size, const float *coors_range, int32_t *grid) {
    for (int j = 0; j < 3; ++j)
        grid[j] = (int32_t)ceilf((coors_range[3 + j] - coors_range[j]) / voxel_size[j]);
}

/* points: [n, stride] fp32 (x, y, z, ...); coors: [n, 3] int32 written as (z, y, x) */
void geomae_oracle_dynamic_voxelize(const float *points, int64_t n, int stride,
                                    const float *voxel_size, const float *coors_range,
                                    int32_t *coors) {
    int32_t grid[3];
    geomae_oracle_grid_size(voxel_size, coors_range, grid);
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < 3; ++j) {
            float q = floorf((points[i * stride + j] - coors_range[j]) / voxel_size[j]);
            int32_t c;
            if (!(q >= 0.0f)) c = 0;                         /* negative or NaN */
            else if (q >= (float)grid[j]) c = grid[j] - 1;
            else c = (int32_t)q;
            coors[i * 3 + (2 - j)] = c;
        }
    }
}
