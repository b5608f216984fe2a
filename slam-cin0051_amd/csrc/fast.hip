// fast.hip -- FAST-N segment test, raster-order compaction and corner score.
//
// Reference behaviour (paths relative to the reference root):
//   detectFASTKeypoints   src/frontend/feature_detector.cpp:56-68   raster scan, border 3
//   isFASTCorner          src/frontend/feature_detector.cpp:70-145  two pretests + 32-step walk
//   computeFASTScore      src/frontend/feature_detector.cpp:190-203 sum of 16 |I_k - I_c|
//   M_PIXEL_OFFSETS       include/slam/frontend/feature_detector.hpp:138-153
//
// MI355X design: integer-only, one image read.
//   K1  fast_detect : one workgroup = 256 x 32 pixel tile, staged through LDS with a 3-px halo as
//                     aligned dword loads; a cardinal quick reject on 4 pixels per lane and instruction
//                     (6-bit bytes, plain 32-bit adds), the segment test (and the score, while the circle is in LDS)
//                     only on the compacted survivors.  Written: the corner predicate of every
//                     64-pixel row segment as one 64-bit word of a corner bitmap (W*H/8 bytes), the
//                     popcount of every word, and the tile's corners as (position, score) entries.
//   K1b scan        : per frame, exclusive prefix sum of the word popcounts in raster order
//                     (row-major, word-major) -> the raster index of every corner is
//                     wordoff[word] + popcount(bits below it).  The generic NMS path uses this
//                     bitmap + prefix pair as its spatial index as well.
//   K1c expand      : a wave per tile list: every entry looks up its raster index and writes
//                     (x,y) and the sort key score<<20|rasterIndex; tiles whose list overflowed
//                     walk their bitmap words and score the corners from the image.
#include "slamx_internal.h"

namespace slamx {

namespace {

// circle offsets {dx,dy}, clockwise from 12 o'clock (feature_detector.hpp:138-153)
// (constexpr, not __constant__: the 16 offsets fold into the immediate fields of the LDS byte reads -- as device
// constants they were fetched and multiplied by the row stride with ~100 scalar instructions per tested pixel)
constexpr int CIRC_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
constexpr int CIRC_DY[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};

constexpr int TILE_W = FAST_TILE_W;
constexpr int TILE_H = FAST_TILE_H;
constexpr int HALO = 3;
constexpr int LDS_ROW_DW = TILE_W / 4 + 2;   // 264 bytes: columns [x0-4, x0+260)
constexpr int LDS_ROWS = TILE_H + 2 * HALO;  // 38
constexpr int ROWS_PER_WAVE = TILE_H / 4;
// survivors of the pretest listed per tile (about 100 of 8192 pixels on the bench frames); a tile with
// more keeps them in the lanes' own flag registers instead (slower, any count) -- the short list
// is what lets 8 workgroups share a CU's LDS
constexpr int SURVIVOR_CAP = 2048;

// The segment test of isFASTCorner on brighter/darker bit masks of the 16 circle pixels; circle pixel q
// is bit 15 - q (the masks are built by shifting sign bits in), which a circular run does not mind.
__device__ __forceinline__ bool segment_test(uint32_t B, uint32_t D, int arc) {
    // pretest A (feature_detector.cpp:81-96): pixels 0 and 8
    if (((B | D) & 0x8080u) == 0) {
        return false;
    }
    // pretest B (:98-113): pixels 0,4,8,12 -- applied for every arc length, as the reference does
    if (__popc(B & 0x8888u) < 3 && __popc(D & 0x8888u) < 3) {
        return false;
    }
    // 32-step circular walk (:118-142): true iff the doubled sequence has a run >= arc.
    if (arc <= 0) {
        return true;  // "0 >= 0" fires on the first step
    }
    // bit i of rb / rd: the doubled sequence holds a run of `len` starting at i; runs double, the rest of the
    // arc is one more overlapping step (arc <= 16, at most five steps instead of arc - 1)
    uint32_t rb = B | (B << 16);
    uint32_t rd = D | (D << 16);
    int len = 1;
    while (2 * len <= arc) {
        rb &= rb >> len;
        rd &= rd >> len;
        len *= 2;
    }
    const int rest = arc - len;  // < len
    rb &= rb >> rest;
    rd &= rd >> rest;
    return (rb | rd) != 0;
}

// Quick reject on FOUR pixels per register (a superset of the reference's pretest B, feature_detector.cpp:
// 98-113; pretest A, :81-96, is implied -- 3 of 4 cardinals include pixel 0 or pixel 8).  Pixels are
// reduced to 6 bits (p >> 2) so that every byte keeps two guard bits and whole dwords can be compared
// with plain 32-bit adds and subtracts, which issue faster than any packed or three-operand form
// (profiles/r02_op_rates_microbench.txt).  With t6 = (t + 1) >> 2:
//     n > c + t  =>  (n >> 2) >= (c >> 2) + t6          n < c - t  =>  (n >> 2) <= (c >> 2) - t6
// (floor is monotone and superadditive), so a pixel the reference's pretest lets through always
// passes here; the few extra survivors fail the exact segment test of phase 2.
//   brighter:  n6 + (128 - t6 - c6)   has bit 7 set  <=>  n6 >= c6 + t6      (range 2 .. 191: no carry)
//   darker:    (c6 + 128 - t6) - n6   has bit 7 set  <=>  n6 <= c6 - t6      (range 2 .. 191: no borrow)
constexpr uint32_t SWAR_LOW6 = 0x3f3f3f3fu;
constexpr uint32_t SWAR_FLAG = 0x80808080u;

__device__ __forceinline__ uint32_t to6(uint32_t w) { return (w >> 2) & SWAR_LOW6; }

// bit 7 of every byte: at least 3 of the 4 words a, b, c, d have it set
__device__ __forceinline__ uint32_t three_of_four(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t maj = __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8);   // at least two of a, b, c
    const uint32_t all3 = __builtin_amdgcn_bitop3_b32(a, b, c, 0x80);  // a & b & c
    return __builtin_amdgcn_bitop3_b32(all3, d, maj, 0xf8);            // all3 | (d & maj)
}

// c6: centre pixels, u / d / l / r: the cardinal circle pixels 0, 8, 12, 4 of the same four centres, all as
// 6-bit bytes; gt = 128 - t6 in every byte.  Bit 7 of byte i of the result = pixel i passes.
__device__ __forceinline__ uint32_t cardinal_test4(uint32_t c6, uint32_t u, uint32_t d, uint32_t l, uint32_t r, uint32_t gt) {
    const uint32_t kb = gt - c6;
    const uint32_t kd = c6 + gt;
    const uint32_t bright = three_of_four(u + kb, d + kb, l + kb, r + kb);
    const uint32_t dark = three_of_four(kd - u, kd - d, kd - l, kd - r);
    return bright | dark;
}

// Two phases per 256 x 32 tile (a lane owns 4 horizontally adjacent pixels = one dword).
//  phase 1: a byte-parallel quick reject on 4 pixels per lane and instruction (cardinal_test4; centre
//           row dwords reduced to 6 bits once and reused as the up / down rows of the rows 3 below /
//           above; left / right neighbours with v_alignbyte); about 1 % of the pixels survive and
//           go to an LDS list.
//  phase 2: the 16-pixel segment test on the survivors, on dense waves.
// Pixels that fail the pretest can never be corners, so the bitmap is the reference's.
__global__ __launch_bounds__(256) void fast_detect_kernel(BatchView v, int threshold, int arc, int nframes, int tiles_x, int tiles_y, uint32_t tiles_x_recip) {
    __shared__ uint32_t tile[LDS_ROWS * LDS_ROW_DW];
    __shared__ uint16_t survivors[SURVIVOR_CAP];
    __shared__ uint32_t row_mask[TILE_H][TILE_W / 32];
    __shared__ uint32_t corners[CORNER_LIST_CAP];  // score << 16 | row in tile << 8 | column in tile
    __shared__ int n_survivors;
    __shared__ int n_corners;
    int f, tile_id;
    if (!frame_block(nframes, tiles_x * tiles_y, &f, &tile_id)) {
        return;
    }
    // tile_id / tiles_x as a multiply-high by ceil(2^32 / tiles_x): exact while tile_id * tiles_x < 2^32, and a frame
    // of 65535 x 65535 pixels has 2^19 tiles in rows of 2^8 (0 stands for tiles_x == 1, whose reciprocal does not fit)
    const int ty = tiles_x_recip != 0 ? static_cast<int>(__umulhi(static_cast<uint32_t>(tile_id), tiles_x_recip)) : tile_id;
    const int tx = tile_id - ty * tiles_x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));  // scalar: row addresses on the SALU
    const uint8_t* img = v.frames + static_cast<size_t>(f) * v.frame_stride;
    const int x0 = tx * TILE_W - 4;
    const int y0 = ty * TILE_H - HALO;
    {
        // a wave stages whole rows (lane -> dword column), threads 0..2*LDS_ROWS-1 the two extra
        // halo dwords of every row; all loads are issued before the first LDS store
        constexpr int NLOAD = (LDS_ROWS + 3) / 4;
        // Buffer loads: the row offset travels in a scalar register and the column offset in the lane, so a load
        // costs no vector address math, and what lies outside the frame is zero-filled by the range check
        // instead of selects: lanes left / right of the frame carry an offset beyond every buffer, rows above /
        // below it go through a descriptor of 0 bytes.
        const int frame_bytes = v.rows * v.pitch;
        const int xa = x0 + lane * 4;
        const uint32_t xoff = xa >= 0 && xa + 4 <= v.pitch ? static_cast<uint32_t>(xa) : 0xffffffffu;
        uint32_t va[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int r = wave + 4 * i;  // wave-uniform
            const int yy = y0 + r;
            const bool row_ok = r < LDS_ROWS && yy >= 0 && yy < v.rows;
            const __amdgpu_buffer_rsrc_t rows_of_frame =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, row_ok ? frame_bytes : 0, 0x00020000);
            va[i] = __builtin_amdgcn_raw_buffer_load_b32(rows_of_frame, xoff, (row_ok ? yy : 0) * v.pitch, 0);
        }
        const int hr = static_cast<int>(threadIdx.x) >> 1;
        const int hx = x0 + (64 + (static_cast<int>(threadIdx.x) & 1)) * 4;
        const int hy = y0 + hr;
        const bool hok = hr < LDS_ROWS && hy >= 0 && hy < v.rows && hx + 4 <= v.pitch;
        const __amdgpu_buffer_rsrc_t frame = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, frame_bytes, 0x00020000);
        const uint32_t vh = __builtin_amdgcn_raw_buffer_load_b32(frame, hok ? static_cast<uint32_t>(hy * v.pitch + hx) : 0xffffffffu, 0, 0);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int r = wave + 4 * i;
            if (r < LDS_ROWS) {
                tile[r * LDS_ROW_DW + lane] = va[i];
            }
        }
        if (hr < LDS_ROWS) {
            tile[hr * LDS_ROW_DW + 64 + (threadIdx.x & 1)] = vh;
        }
    }
    if (threadIdx.x < TILE_H * (TILE_W / 32)) {
        row_mask[threadIdx.x / (TILE_W / 32)][threadIdx.x % (TILE_W / 32)] = 0;
    }
    if (threadIdx.x == 0) {
        n_survivors = 0;
        n_corners = 0;
    }
    __syncthreads();
    const int x = tx * TILE_W + lane * 4;
    // ---- phase 1: a wave walks its ROWS_PER_WAVE consecutive rows; every LDS row is reduced to 6-bit
    // bytes once and serves as the centre row, as the "up" row 3 rows later and the "down" row 3 rows
    // earlier.  Pass flags are collected as bits (pixel i of row k -> bit 8 i + k) and compacted once
    // at the end.
    uint32_t acc = 0;
    {
        const int t6 = min((threshold + 1) >> 2, 63);
        uint32_t gt = SWAR_FLAG - static_cast<uint32_t>(t6) * 0x01010101u;
        // a vector register: the fast 32-bit forms lose their issue rate with a scalar operand
        asm volatile("v_mov_b32 %0, %0" : "+v"(gt));
        const uint32_t* col = tile + (wave * ROWS_PER_WAVE) * LDS_ROW_DW + lane + 1;  // LDS row of tile row ry-3
        uint32_t c6[ROWS_PER_WAVE + 2 * HALO];
#pragma unroll
        for (int j = 0; j < 2 * HALO; ++j) {
            c6[j] = to6(col[j * LDS_ROW_DW]);
        }
#pragma unroll
        for (int k = 0; k < ROWS_PER_WAVE; ++k) {
            const int jn = k + 2 * HALO;
            c6[jn] = to6(col[jn * LDS_ROW_DW]);
            const int jc = k + HALO;
            const uint32_t C = c6[jc];
            const uint32_t Pv = to6(col[jc * LDS_ROW_DW - 1]);
            const uint32_t Nx = to6(col[jc * LDS_ROW_DW + 1]);
            const uint32_t l = __builtin_amdgcn_alignbyte(C, Pv, 1);   // pixels x-3 .. x
            const uint32_t r = __builtin_amdgcn_alignbyte(Nx, C, 3);   // pixels x+3 .. x+6
            const uint32_t pass = cardinal_test4(C, c6[k], c6[jn], l, r, gt);
            acc = (acc >> 1) | (pass & SWAR_FLAG);  // after the last row: row k at bit k of every byte
        }
    }
    // image border: rows / columns < 3 or >= size - 3 are never tested (feature_detector.cpp:61-63); most tiles
    // do not touch it (block-uniform branch)
    if (tx * TILE_W < HALO || (tx + 1) * TILE_W > v.cols - HALO || ty * TILE_H < HALO || (ty + 1) * TILE_H > v.rows - HALO) {
        const int ybase = ty * TILE_H + wave * ROWS_PER_WAVE;
        uint32_t rowbits = 0;
#pragma unroll
        for (int k = 0; k < ROWS_PER_WAVE; ++k) {
            if (ybase + k >= HALO && ybase + k < v.rows - HALO) {
                rowbits |= 1u << k;
            }
        }
        const uint32_t ok0 = (x + 0 >= HALO && x + 0 < v.cols - HALO) ? rowbits : 0u;
        const uint32_t ok1 = (x + 1 >= HALO && x + 1 < v.cols - HALO) ? rowbits : 0u;
        const uint32_t ok2 = (x + 2 >= HALO && x + 2 < v.cols - HALO) ? rowbits : 0u;
        const uint32_t ok3 = (x + 3 >= HALO && x + 3 < v.cols - HALO) ? rowbits : 0u;
        acc &= ok0 | (ok1 << 8) | (ok2 << 16) | (ok3 << 24);
    }
    {
        const int cnt = __popc(acc);
        if (cnt != 0) {
            // entry = thread << 5 | flag bit (13 bits), decoded by phase 2.  A lane whose entries do not all
            // fit skips the list: the total then exceeds the capacity and phase 2 takes the dense path.
            const int pos = atomicAdd(&n_survivors, cnt);
            if (pos + cnt <= SURVIVOR_CAP) {
                uint16_t* dst = survivors + pos;
                const uint32_t ent0 = threadIdx.x << 5;
                uint32_t m = acc;
                do {
                    *dst++ = static_cast<uint16_t>(ent0 + static_cast<uint32_t>(__ffs(static_cast<int>(m)) - 1));
                    m &= m - 1;
                } while (m != 0);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: the 16-pixel segment test
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile);
    constexpr int RS = LDS_ROW_DW * 4;  // LDS row stride in bytes
    auto test_pixel = [&](int ry, int col) {
        const uint8_t* p = t8 + (ry + HALO) * RS + (col + 4);
        const int c = p[0];
        const int hi = c + threshold;
        const int lo = c - threshold;
        uint32_t B = 0;
        uint32_t D = 0;
        uint32_t score = 0;  // computeFASTScore: sum of |circle pixel - centre| (feature_detector.cpp:190-203)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = p[CIRC_DY[q] * RS + CIRC_DX[q]];
            // shift the sign of (hi - n) / (n - lo) in from the right: no compare, no select
            B = __builtin_amdgcn_alignbit(B, static_cast<uint32_t>(hi - n), 31);
            D = __builtin_amdgcn_alignbit(D, static_cast<uint32_t>(n - lo), 31);
            score = __builtin_amdgcn_sad_u16(static_cast<uint32_t>(n), static_cast<uint32_t>(c), score);
        }
        if (segment_test(B, D, arc)) {
            atomicOr(&row_mask[ry][col >> 5], 1u << (col & 31));
            const int slot = atomicAdd(&n_corners, 1);
            if (slot < CORNER_LIST_CAP) {
                corners[slot] = (score << 16) | static_cast<uint32_t>((ry << 8) | col);
            }
        }
    };
    const int total = n_survivors;
    if (total <= SURVIVOR_CAP) {  // block-uniform
        for (int s = threadIdx.x; s < total; s += 256) {
            const int e = survivors[s];
            test_pixel(((e >> 11) * ROWS_PER_WAVE) + (e & 7), ((e >> 5) & 63) * 4 + ((e >> 3) & 3));
        }
    } else {
        // dense tile: every lane tests the pixels it flagged itself
        while (acc != 0) {
            const int b = __ffs(static_cast<int>(acc)) - 1;
            acc &= acc - 1;
            test_pixel(wave * ROWS_PER_WAVE + (b & 7), lane * 4 + (b >> 3));
        }
    }
    __syncthreads();
    // one 64-bit bitmap word per (row, 64-column group): TILE_H rows x 4 words per tile
    if (threadIdx.x < TILE_H * (TILE_W / 64)) {
        const int ry = threadIdx.x / (TILE_W / 64);
        const int wq = threadIdx.x % (TILE_W / 64);
        const int y = ty * TILE_H + ry;
        const int wx = tx * (TILE_W / 64) + wq;
        if (y < v.rows && wx < v.nw) {
            const size_t w = static_cast<size_t>(f) * v.rows * v.nw + static_cast<size_t>(y) * v.nw + wx;
            const uint32_t m_lo = row_mask[ry][2 * wq], m_hi = row_mask[ry][2 * wq + 1];
            v.masks[w] = static_cast<unsigned long long>(m_lo) | (static_cast<unsigned long long>(m_hi) << 32);
            v.wordcnt[w] = static_cast<uint8_t>(__popc(m_lo) + __popc(m_hi));
        }
    }
    // the tile's corners with their scores, for expand_kernel (which falls back to the bitmap and the
    // image when a tile holds more than the list does)
    const int nc = n_corners;
    const size_t tile_slot = static_cast<size_t>(f) * v.fast_tiles + tile_id;
    if (threadIdx.x == 0) {
        v.tile_count[tile_slot] = static_cast<uint32_t>(nc);
    }
    if (nc <= CORNER_LIST_CAP) {
        for (int i = threadIdx.x; i < nc; i += 256) {
            v.tile_list[tile_slot * CORNER_LIST_CAP + i] = corners[i];
        }
    }
}

// One workgroup per frame: exclusive scan of the per-word corner counts over rows*nw words.  A thread owns four
// consecutive words per step; the running total lives in a register of every thread (all of them add up
// the same 16 wave sums), so a step needs one barrier (the wave sums are double-buffered).
__global__ __launch_bounds__(1024) void scan_kernel(BatchView v) {
    __shared__ uint32_t wave_sums[2][16];
    const int f = blockIdx.x;
    const int n = v.rows * v.nw;
    const uint8_t* __restrict__ cnt = v.wordcnt + static_cast<size_t>(f) * n;
    uint32_t* __restrict__ off = v.wordoff + static_cast<size_t>(f) * n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    constexpr int PER = 4;
    uint32_t carry = 0;  // corners before `base`, identical in all threads
    int buf = 0;
    for (int base = 0; base < n; base += 1024 * PER) {
        const int i0 = base + static_cast<int>(threadIdx.x) * PER;
        uint32_t c[PER];
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            c[k] = i0 + k < n ? static_cast<uint32_t>(cnt[i0 + k]) : 0u;
            t += c[k];
        }
        uint32_t s = t;  // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = __shfl_up(s, d);
            if (lane >= d) {
                s += u;
            }
        }
        if (lane == 63) {
            wave_sums[buf][wave] = s;
        }
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t ws = wave_sums[buf][w];
            before += w < wave ? ws : 0u;
            total += ws;
        }
        uint32_t run = carry + before + s - t;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (i0 + k < n) {
                off[i0 + k] = run;
            }
            run += c[k];
        }
        carry += total;
        buf ^= 1;
    }
    if (threadIdx.x == 0) {
        int m = static_cast<int>(carry);
        int st = 0;
        if (m > v.maxM) {
            st = STATUS_CAND_OVERFLOW;  // fail loudly on the host; downstream kernels see M = 0
            m = 0;
        }
        v.n_cand[f] = m;
        v.status[f] = st;
        v.n_kp[f] = 0;
    }
}

// computeFASTScore for one corner: the 16 circle pixels come from 7 rows; per row one 12-byte load at
// the dword-aligned address below x-3 and two v_alignbyte give the bytes x-3..x and x+1..x+4, the
// circle bytes are gathered four to a dword with v_perm (unused byte positions hold the centre, whose
// difference is 0) and summed with v_sad_u8.
struct RowBytes {
    uint32_t a;  // pixels x-3, x-2, x-1, x
    uint32_t b;  // pixels x+1, x+2, x+3, x+4
};
__device__ __forceinline__ RowBytes load_row7(const uint8_t* row, int x) {
    const int a0 = (x - 3) & ~3;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(row + a0);
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    const uint32_t sh = static_cast<uint32_t>(x - 3 - a0);
    RowBytes r;
    r.a = __builtin_amdgcn_alignbyte(w1, w0, sh);
    r.b = __builtin_amdgcn_alignbyte(w2, w1, sh);
    return r;
}
// v_perm_b32(hi, lo, sel): selector byte 0-3 picks lo.b0-3, 4-7 picks hi.b0-3
__device__ __forceinline__ int corner_score(const uint8_t* img, int pitch, int x, int y) {
    const uint8_t* centre_row = img + static_cast<size_t>(y) * pitch;
    const RowBytes m3 = load_row7(centre_row - 3 * pitch, x);
    const RowBytes m2 = load_row7(centre_row - 2 * pitch, x);
    const RowBytes m1 = load_row7(centre_row - pitch, x);
    const RowBytes z0 = load_row7(centre_row, x);
    const RowBytes p1 = load_row7(centre_row + pitch, x);
    const RowBytes p2 = load_row7(centre_row + 2 * pitch, x);
    const RowBytes p3 = load_row7(centre_row + 3 * pitch, x);
    const uint32_t c4 = __builtin_amdgcn_perm(0u, z0.a, 0x03030303u);  // centre in all four bytes
    // rows -3 / +3: dx -1, 0, +1 = a.b2, a.b3, b.b0; the fourth byte is the centre
    uint32_t s = 0;
    uint32_t t;
    t = __builtin_amdgcn_perm(m3.b, m3.a, 0x00040302u);  // x-1, x, x+1, (replaced below)
    t = (t & 0x00ffffffu) | (c4 & 0xff000000u);
    s = __builtin_amdgcn_sad_u8(t, c4, s);
    t = __builtin_amdgcn_perm(p3.b, p3.a, 0x00040302u);
    t = (t & 0x00ffffffu) | (c4 & 0xff000000u);
    s = __builtin_amdgcn_sad_u8(t, c4, s);
    // rows -2 / +2: dx -2, +2 = a.b1, b.b1 ; rows -1 / +1: dx -3, +3 = a.b0, b.b2
    t = __builtin_amdgcn_perm(m2.b, m2.a, 0x00000501u);  // bytes: x-2, x+2, (2 fillers)
    uint32_t u = __builtin_amdgcn_perm(m1.b, m1.a, 0x06000000u);  // bytes 2,3: x-3 (a.b0), x+3 (b.b2)
    t = (t & 0x0000ffffu) | (u & 0xffff0000u);
    s = __builtin_amdgcn_sad_u8(t, c4, s);
    t = __builtin_amdgcn_perm(p2.b, p2.a, 0x00000501u);
    u = __builtin_amdgcn_perm(p1.b, p1.a, 0x06000000u);
    t = (t & 0x0000ffffu) | (u & 0xffff0000u);
    s = __builtin_amdgcn_sad_u8(t, c4, s);
    // row 0: dx -3, +3 = a.b0, b.b2
    t = __builtin_amdgcn_perm(z0.b, z0.a, 0x00060000u);  // bytes: x-3, (filler), x+3, (filler)
    t = (t & 0x00ff00ffu) | (c4 & 0xff00ff00u);
    s = __builtin_amdgcn_sad_u8(t, c4, s);
    return static_cast<int>(s);
}

// Corners -> (x,y) and the sort key score<<20 | rasterIndex, in raster order.  A wave walks 64
// bitmap words at a time (coalesced 8-byte loads; about a quarter of the words hold corners, in
// clusters), every lane peels off the set bits of its word and the wave compacts them into an LDS
// list, so that the score of 64 corners at a time is computed on full waves.
__global__ __launch_bounds__(256) void expand_kernel(BatchView v, int nframes, int blocks_per_frame, int tiles_x) {
    __shared__ uint2 list[4][128];  // per wave: {y << 16 | x, raster index} (bitmap path only)
    int f, bx;
    if (!frame_block(nframes, blocks_per_frame, &f, &bx)) {
        return;
    }
    const int m_total = v.n_cand[f];
    if (m_total == 0) {
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int n = v.rows * v.nw;
    const unsigned long long* __restrict__ masks = v.masks + static_cast<size_t>(f) * n;
    const uint32_t* __restrict__ off = v.wordoff + static_cast<size_t>(f) * n;
    const uint8_t* __restrict__ img = v.frames + static_cast<size_t>(f) * v.frame_stride;
    const uint32_t* __restrict__ tcount = v.tile_count + static_cast<size_t>(f) * v.fast_tiles;
    const uint32_t* __restrict__ tlist = v.tile_list + static_cast<size_t>(f) * v.fast_tiles * CORNER_LIST_CAP;
    uint32_t* xy = v.cand_xy + static_cast<size_t>(f) * v.maxM;
    uint32_t* keys = v.keys + static_cast<size_t>(f) * v.maxM;
    uint2* q = list[wave];
    for (int t = bx * 4 + wave; t < v.fast_tiles; t += blocks_per_frame * 4) {  // wave-uniform
        const int nc = __builtin_amdgcn_readfirstlane(static_cast<int>(tcount[t]));
        if (nc == 0) {
            continue;
        }
        const int ty = t / tiles_x;
        const int tx = t - ty * tiles_x;
        if (nc <= CORNER_LIST_CAP) {
            // the detector left position and score of every corner of the tile: the raster index is the
            // word's prefix plus the corners before it in its bitmap word
            for (int base = 0; base < nc; base += 64) {
                const int i = base + lane;
                if (i < nc) {
                    const uint32_t e = tlist[static_cast<size_t>(t) * CORNER_LIST_CAP + i];
                    const int x = tx * FAST_TILE_W + static_cast<int>(e & 0xffu);
                    const int y = ty * FAST_TILE_H + static_cast<int>((e >> 8) & 0xffu);
                    const int w = y * v.nw + (x >> 6);
                    const unsigned long long below = masks[w] & ((1ull << (x & 63)) - 1ull);
                    const uint32_t idx = off[w] + static_cast<uint32_t>(__popcll(below));
                    xy[idx] = (static_cast<uint32_t>(y) << 16) | static_cast<uint32_t>(x);
                    keys[idx] = ((e >> 16) << SCORE_SHIFT) | idx;
                }
            }
            continue;
        }
        // crowded tile: walk its bitmap words and score the corners from the image
        int cnt = 0;  // entries waiting in q (wave-uniform)
        auto emit = [&](int count) {  // the first `count` (<= 64) entries of q
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < count) {
                const uint2 e = q[lane];
                const int x = static_cast<int>(e.x & 0xffffu);
                const int y = static_cast<int>(e.x >> 16);
                const int score = corner_score(img, v.pitch, x, y);
                xy[e.y] = e.x;
                keys[e.y] = (static_cast<uint32_t>(score) << SCORE_SHIFT) | e.y;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        constexpr int WPR = FAST_TILE_W / 64;  // bitmap words per tile row
        for (int wbase = 0; wbase < FAST_TILE_H * WPR; wbase += 64) {
            const int wi = wbase + lane;
            const int row = ty * FAST_TILE_H + wi / WPR;
            const int wx = tx * WPR + wi % WPR;
            const bool inside = row < v.rows && wx < v.nw;
            const int w = row * v.nw + wx;
            unsigned long long m = inside ? masks[w] : 0ull;
            const int x0 = wx * 64;
            uint32_t idx = m != 0 ? off[w] : 0u;
            while (true) {
                const bool has = m != 0;
                const unsigned long long bal = __ballot(has);
                if (bal == 0) {
                    break;
                }
                if (has) {
                    const int b = __ffsll(static_cast<long long>(m)) - 1;
                    m &= m - 1;
                    const int slot = cnt + static_cast<int>(__builtin_amdgcn_mbcnt_hi(
                                               static_cast<uint32_t>(bal >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(bal), 0u)));
                    q[slot] = make_uint2((static_cast<uint32_t>(row) << 16) | static_cast<uint32_t>(x0 + b), idx);
                    ++idx;
                }
                cnt += __popcll(bal);
                if (cnt >= 64) {
                    emit(64);
                    cnt -= 64;
                    if (lane < cnt) {  // move the tail to the front
                        const uint2 e = q[64 + lane];
                        q[lane] = e;
                    }
                }
            }
        }
        emit(cnt);
    }
}

}  // namespace

void launch_fast_detect(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream) {
    const int tiles_x = (v.cols + TILE_W - 1) / TILE_W;
    const int tiles_y = (v.rows + TILE_H - 1) / TILE_H;
    const uint32_t tiles_x_recip = tiles_x > 1 ? 0xffffffffu / static_cast<uint32_t>(tiles_x) + 1u : 0u;
    hipLaunchKernelGGL(fast_detect_kernel, frame_grid(nframes, tiles_x * tiles_y), dim3(256), 0, stream, v, p.threshold, p.arc,
                       nframes, tiles_x, tiles_y, tiles_x_recip);
}

void launch_scan(const BatchView& v, int nframes, hipStream_t stream) {
    hipLaunchKernelGGL(scan_kernel, dim3(nframes), dim3(1024), 0, stream, v);
}

void launch_expand(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream) {
    (void)p;
    const int tiles_x = (v.cols + TILE_W - 1) / TILE_W;
    const int bpf = grid_x(v.fast_tiles, 4, nframes);
    hipLaunchKernelGGL(expand_kernel, dim3(frame_grid(nframes, bpf)), dim3(256), 0, stream, v, nframes, bpf, tiles_x);
}

}  // namespace slamx
