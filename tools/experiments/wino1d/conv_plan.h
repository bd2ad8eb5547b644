// conv_plan.h -- how an image or a sub-rectangle of hr x wr pixels is cut into blocks, for the direct split-f16 kernel
// (conv_split.hip: M tiles of 32 pixels, at most 8 per block) and for its Winograd form (conv_wino.hip: M tiles of 32 pixel
// PAIRS, at most 4 per block).  Shared by the kernels' launch code and by k_rect_plan, which runs on the device.
#ifndef SNK_CONV_PLAN_H
#define SNK_CONV_PLAN_H

#define HS_LDP 80                                        // bytes per LDS pixel: [hi k0-15 | lo k0-15] + 16 (16 x odd: conflict-free b128)
#define HS_NPB 352                                       // pixels per LDS buffer
#define HS_NST 5                                         // staging items per thread and chunk: 64 pixels x 4 float4 each
#define HW_NPB 176                                       // conv_wino.hip: pair positions per LDS plane
#define HW_NMAX 3                                        //   M tiles per block (4 would need 8 more accumulator tiles than the register file has: measured, spills)

// parts (blocks) a rectangle of hr x wr pixels is cut into and the M tiles of its largest part: the fewest parts of at most
// 8 tiles whose strip (their rows, one above, one below, halo columns) fits the LDS buffer and the staging items
__host__ __device__ static inline int hs_rect_parts(int hr, int wr, int Hd, int Wd, int *tiles_max)
{
    const int T = (hr * wr + 31) / 32;
    for (int parts = (T + 7) / 8;; ++parts) {
        const int tm = (T + parts - 1) / parts;
        int rows_out = (tm * 32 + wr - 2) / wr + 1;                       // worst alignment of 32 tm pixels
        if (rows_out > hr) rows_out = hr;
        const int rows_in = rows_out + 2 < Hd ? rows_out + 2 : Hd, cols_in = wr + 2 < Wd ? wr + 2 : Wd;
        const bool fits = (rows_out + 2) * (wr + 2) <= HS_NPB && rows_in * cols_in <= 64 * HS_NST;
        if (fits || tm == 1) { *tiles_max = tm; return fits ? parts : -1; }
    }
}


// the same for the Winograd kernel: tiles of 32 pairs (wp = (wr + 1) / 2 pairs per row), at most HW_NMAX per part, whose strip
// (their pair rows, one above, one below) fits an LDS plane and the staging items
__host__ __device__ static inline int hw_rect_parts(int hr, int wr, int *tiles_max)
{
    const int wp = (wr + 1) / 2, T = (hr * wp + 31) / 32;
    for (int parts = (T + HW_NMAX - 1) / HW_NMAX;; ++parts) {
        const int tm = (T + parts - 1) / parts;
        int rows_out = (tm * 32 + wp - 2) / wp + 1;                       // worst alignment of 32 tm pairs
        if (rows_out > hr) rows_out = hr;
        const bool fits = (rows_out + 2) * wp <= HW_NPB && (rows_out + 2) * ((wp + 1) / 2) <= 128;      // LDS plane; one staging item (two pairs x 4 channels) per thread
        if (fits || tm == 1) { *tiles_max = tm; return fits ? parts : -1; }
    }
}
#endif
