// Lewiner MC33 marching cubes on gfx950, bit-compatible with
//     skimage.measure.marching_cubes_lewiner(vol, level)            (scikit-image 0.17.2 / 0.18.3 Cython core)
// as called by the reference at /root/reference/lib/mesh_util.py:40,45.
//
// The Cython core is a sequential sweep (axis 0 outer, axis 2 inner) with a two-layer vertex cache; vertex ids
// are handed out in order of first use.  Facts that make a parallel restatement exact (SURVEY.md A.7, and
// pinned by tests/golden/mc_*):
//   * every cell that contains an intersected lattice edge references it, so the vertex of an edge is created
//     by the FIRST cell (in sweep order) containing it: the cell "below/left/behind".  Interior cells therefore
//     own only the three edges meeting at their far corner (5, 6, 10) and their centre vertex (12);
//   * inside the owning cell vertices are numbered in order of first appearance in its triangle list;
//   * faces are emitted cell by cell in sweep order, triangles in LUT order.
// So:  pass 1  classify: MC33 tests on every cell that the surface crosses; the chosen triangle list is stored as a 32-bit
//              cell code (table, offset, triangle count, vertices created; 0 = empty cell), and vertices / triangles /
//              active cells are summed per block of 1024 sweep-consecutive cells
//      pass 2  exclusive scan of the block sums
//      pass 3  compaction: the active cells in sweep order, found with wavefront ballots over the cell codes; the
//              in-block prefixes of their vertex and triangle counts are sums of popcounts of per-bit ballots.
//              One 16-byte entry per active cell: (cell, code, first vertex id, first triangle id)
//      pass 4a vertices, one thread per ACTIVE cell: positions; ids stored in 4 dense per-voxel tables
//              (x-edge, y-edge, z-edge starting at the voxel, centre of the cell whose corner 0 it is)
//      pass 4b faces (+ values by atomic max, normals by atomic add): vertex ids looked up in the tables
//      pass 5  normalise normals.
// HBM-bound: the volume is read once (pass 1; corner re-reads hit L1/L2), then 4 bytes per cell (pass 3) and the
// active cells only; the ambiguity tests run in double, once, on the active cells.  All arithmetic that decides topology or positions is double, written exactly
// as the Cython core evaluates it (compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "surs_common.h"

namespace surs {
namespace mc {

#define MCL_QUAL __device__ const
#include "mc_luts.inc"
#undef MCL_QUAL

// the Cython core defines FLT_EPSILON = np.spacing(1.0): DOUBLE epsilon (black-box verified, see oracle)
#define MC_EPS 2.220446049250313e-16

constexpr int CELLS_PER_BLOCK = 1024;
constexpr int THREADS = 256;

struct Dims {
    int nz, ny, nx;     // volume dims (axis 0, 1, 2)
    int cz, cy, cx;     // cells per axis
    long long ncells;
    long long cell_begin, cell_end;   // the flat (sweep-order) cell range this call processes
    int base_verts, base_faces;      // vertices / triangles produced by earlier ranges
    int zoff;                        // slab mode: index of the volume's plane 0 in the whole grid (0 = the grid's bottom)
};

struct Tiling {
    const signed char *row;  // 3 * nt edge numbers (12 = the centre vertex)
    int nt;
    int tab, off;            // which triangle table and the element offset of `row` in it: what the cell code stores
};

// every triangle table of the MC33 look-up tables, so that a tiling can be stored as (table, offset) in 32 bits
#define MC_TILING_TABLES(X) \
    X(TILING1) \
    X(TILING10_1_1) \
    X(TILING10_1_1_) \
    X(TILING10_1_2) \
    X(TILING10_2) \
    X(TILING10_2_) \
    X(TILING11) \
    X(TILING12_1_1) \
    X(TILING12_1_1_) \
    X(TILING12_1_2) \
    X(TILING12_2) \
    X(TILING12_2_) \
    X(TILING13_1) \
    X(TILING13_1_) \
    X(TILING13_2) \
    X(TILING13_2_) \
    X(TILING13_3) \
    X(TILING13_3_) \
    X(TILING13_4) \
    X(TILING13_5_1) \
    X(TILING13_5_2) \
    X(TILING14) \
    X(TILING2) \
    X(TILING3_1) \
    X(TILING3_2) \
    X(TILING4_1) \
    X(TILING4_2) \
    X(TILING5) \
    X(TILING6_1_1) \
    X(TILING6_1_2) \
    X(TILING6_2) \
    X(TILING7_1) \
    X(TILING7_2) \
    X(TILING7_3) \
    X(TILING7_4_1) \
    X(TILING7_4_2) \
    X(TILING8) \
    X(TILING9)
enum {
#define X(name) TAB_##name,
    MC_TILING_TABLES(X)
#undef X
        TAB_COUNT
};
__device__ const signed char *const MC_TAB_PTR[TAB_COUNT] = {
#define X(name) MCL_##name,
    MC_TILING_TABLES(X)
#undef X
};

#define P1(name, cfg) (MCL_##name + (size_t)(cfg) * MCL_##name##_D1)   /* row of a TEST table */
#define P2(name, cfg, sub) (MCL_##name + ((size_t)(cfg) * MCL_##name##_D1 + (sub)) * MCL_##name##_D2)
#define T1(name, cfg, ntri) mk_tiling(TAB_##name, MCL_##name, (int)((cfg) * MCL_##name##_D1), ntri)
#define T2(name, cfg, sub, ntri) mk_tiling(TAB_##name, MCL_##name, (int)(((cfg) * MCL_##name##_D1 + (sub)) * MCL_##name##_D2), ntri)
__device__ __forceinline__ Tiling mk_tiling(int tab, const signed char *base, int off, int nt) {
    Tiling t;
    t.row = base + off;
    t.nt = nt;
    t.tab = tab;
    t.off = off;
    return t;
}

// cell code (uint32, 0 = no surface in the cell): bits 0-3 triangles (1..12), 4-7 vertices the cell creates (0..13: a
// corner cell of the volume owns all 12 edges and the centre), 8-13 table, 14-31 offset
__device__ __forceinline__ unsigned encode_cell(const Tiling &t, int nv) {
    return (unsigned)t.nt | ((unsigned)nv << 4) | ((unsigned)t.tab << 8) | ((unsigned)t.off << 14);
}
__device__ __forceinline__ int code_nt(unsigned code) { return (int)(code & 15u); }
__device__ __forceinline__ int code_nv(unsigned code) { return (int)((code >> 4) & 15u); }
__device__ __forceinline__ Tiling decode_cell(unsigned code) {
    Tiling t;
    t.nt = code_nt(code);
    t.tab = (int)((code >> 8) & 63u);
    t.off = (int)(code >> 14);
    t.row = MC_TAB_PTR[t.tab] + t.off;
    return t;
}

__device__ __forceinline__ bool test_face(const double *v, int face) {
    double A = 0, B = 0, C = 0, D = 0;
    switch (face < 0 ? -face : face) {
        case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
        case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
        case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
        case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
        case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
        case 6: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break;
        default: break;
    }
    const double t = A * C - B * D;
    if (t > -MC_EPS && t < MC_EPS) return face >= 0;
    return (double)face * A * t >= 0;
}

__device__ const signed char MC_TI_EDGES[12][8] = {
    {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
    {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
    {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};

__device__ __forceinline__ bool test_internal(const double *v, int mccase, int config, int subconfig, int s) {
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0;
    if (mccase == 4 || mccase == 10) {
        const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        int edge = -1;
        if (mccase == 6) edge = P1(TEST6, config)[2];
        else if (mccase == 7) edge = P1(TEST7, config)[4];
        else if (mccase == 12) edge = P1(TEST12, config)[3];
        else if (mccase == 13) edge = P2(TILING13_5_1, config, subconfig)[0];
        if (edge < 0 || edge > 11) return s < 0;
        const signed char *e = MC_TI_EDGES[edge];
        t = v[e[0]] / (v[e[0]] - v[e[1]] + MC_EPS);
        At = 0;
        Bt = v[e[2]] + (v[e[3]] - v[e[2]]) * t;
        Ct = v[e[4]] + (v[e[5]] - v[e[4]]) * t;
        Dt = v[e[6]] + (v[e[7]] - v[e[6]]) * t;
    }
    int test = 0;
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {
        case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
        case 5: if (At * Ct - Bt * Dt < MC_EPS) return s > 0; break;
        case 10: if (At * Ct - Bt * Dt >= MC_EPS) return s > 0; break;
        case 7: case 11: case 13: case 14: case 15: return s < 0;
        default: break;
    }
    return false;  // the Cython function falls off its end here (returns 0); Lewiner's C++ returns s < 0
}

// MC33 case dispatch (Lewiner's process_cube): which triangle list does this cube get?
__device__ __forceinline__ Tiling select_tiling(const double *v, int index) {
    Tiling r;
    r.row = nullptr;
    r.nt = 0;
    r.tab = 0;
    r.off = 0;
    const int mccase = MCL_CASES[2 * index], config = MCL_CASES[2 * index + 1];
    int sub = 0;
    switch (mccase) {
        case 1: r = T1(TILING1, config, 1); break;
        case 2: r = T1(TILING2, config, 2); break;
        case 3:
            if (test_face(v, MCL_TEST3[config])) { r = T1(TILING3_2, config, 4); }
            else { r = T1(TILING3_1, config, 2); }
            break;
        case 4:
            if (test_internal(v, mccase, config, sub, MCL_TEST4[config])) { r = T1(TILING4_1, config, 2); }
            else { r = T1(TILING4_2, config, 6); }
            break;
        case 5: r = T1(TILING5, config, 3); break;
        case 6:
            if (test_face(v, P1(TEST6, config)[0])) { r = T1(TILING6_2, config, 5); }
            else if (test_internal(v, mccase, config, sub, P1(TEST6, config)[1])) { r = T1(TILING6_1_1, config, 3); }
            else { r = T1(TILING6_1_2, config, 9); }
            break;
        case 7:
            if (test_face(v, P1(TEST7, config)[0])) sub += 1;
            if (test_face(v, P1(TEST7, config)[1])) sub += 2;
            if (test_face(v, P1(TEST7, config)[2])) sub += 4;
            switch (sub) {
                case 0: r = T1(TILING7_1, config, 3); break;
                case 1: r = T2(TILING7_2, config, 0, 5); break;
                case 2: r = T2(TILING7_2, config, 1, 5); break;
                case 3: r = T2(TILING7_3, config, 0, 9); break;
                case 4: r = T2(TILING7_2, config, 2, 5); break;
                case 5: r = T2(TILING7_3, config, 1, 9); break;
                case 6: r = T2(TILING7_3, config, 2, 9); break;
                default:
                    if (test_internal(v, mccase, config, sub, P1(TEST7, config)[3])) { r = T1(TILING7_4_2, config, 9); }
                    else { r = T1(TILING7_4_1, config, 5); }
                    break;
            }
            break;
        case 8: r = T1(TILING8, config, 2); break;
        case 9: r = T1(TILING9, config, 4); break;
        case 10:
            if (test_face(v, P1(TEST10, config)[0])) {
                if (test_face(v, P1(TEST10, config)[1])) { r = T1(TILING10_1_1_, config, 4); }
                else { r = T1(TILING10_2, config, 8); }
            } else {
                if (test_face(v, P1(TEST10, config)[1])) { r = T1(TILING10_2_, config, 8); }
                else if (test_internal(v, mccase, config, sub, P1(TEST10, config)[2])) { r = T1(TILING10_1_1, config, 4); }
                else { r = T1(TILING10_1_2, config, 8); }
            }
            break;
        case 11: r = T1(TILING11, config, 4); break;
        case 12:
            if (test_face(v, P1(TEST12, config)[0])) {
                if (test_face(v, P1(TEST12, config)[1])) { r = T1(TILING12_1_1_, config, 4); }
                else { r = T1(TILING12_2, config, 8); }
            } else {
                if (test_face(v, P1(TEST12, config)[1])) { r = T1(TILING12_2_, config, 8); }
                else if (test_internal(v, mccase, config, sub, P1(TEST12, config)[2])) { r = T1(TILING12_1_1, config, 4); }
                else { r = T1(TILING12_1_2, config, 8); }
            }
            break;
        case 13:
            for (int k = 0; k < 6; ++k)
                if (test_face(v, P1(TEST13, config)[k])) sub += 1 << k;
            sub = MCL_SUBCONFIG13[sub];
            if (sub == 0) { r = T1(TILING13_1, config, 4); }
            else if (sub <= 6) { r = T2(TILING13_2, config, sub - 1, 6); }
            else if (sub <= 18) { r = T2(TILING13_3, config, sub - 7, 10); }
            else if (sub <= 22) { r = T2(TILING13_4, config, sub - 19, 12); }
            else if (sub <= 26) {
                sub -= 23;
                if (test_internal(v, mccase, config, sub, P1(TEST13, config)[6])) { r = T2(TILING13_5_1, config, sub, 6); }
                else { r = T2(TILING13_5_2, config, sub, 10); }
            } else if (sub <= 38) { r = T2(TILING13_3_, config, sub - 27, 10); }
            else if (sub <= 44) { r = T2(TILING13_2_, config, sub - 39, 6); }
            else if (sub == 45) { r = T1(TILING13_1_, config, 4); }
            break;
        case 14: r = T1(TILING14, config, 4); break;
        default: break;
    }
    return r;
}

// which of the 13 "edges" (12 = centre) of cell (x,y,z) does the cell itself create (bit mask)
__device__ __forceinline__ unsigned owned_mask(int x, int y, int z) {
    unsigned m = (1u << 5) | (1u << 6) | (1u << 10) | (1u << 12);
    if (y == 0) m |= (1u << 4) | (1u << 9);
    if (z == 0) m |= (1u << 2) | (1u << 1);
    if (x == 0) m |= (1u << 7) | (1u << 11);
    if (y == 0 && z == 0) m |= 1u << 0;
    if (x == 0 && z == 0) m |= 1u << 3;
    if (x == 0 && y == 0) m |= 1u << 8;
    return m;
}

struct Cell {
    double v[8];
    int index;
};

__device__ __forceinline__ void load_cell(const float *__restrict__ vol, const Dims &d, int x, int y, int z, double level,
                                          Cell &c, float &lo, float &hi) {
    const size_t sy = (size_t)d.nx, sz = (size_t)d.nx * d.ny;
    const float *p = vol + (size_t)z * sz + (size_t)y * sy + x;
    const float f[8] = {p[0], p[1], p[sy + 1], p[sy], p[sz], p[sz + 1], p[sz + sy + 1], p[sz + sy]};
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c.v[k] = (double)f[k] - level;
        if (c.v[k] > 0.0) idx |= 1 << k;
        lo = fminf(lo, f[k]);
        hi = fmaxf(hi, f[k]);
    }
    c.index = idx;
}

// (the cell count is < 2^32, checked by the host: 32-bit unsigned division - the 64-bit one costs more than the rest of
//  the classification of an empty cell)
__device__ __forceinline__ void cell_xyz(const Dims &d, long long c, int &x, int &y, int &z) {
    const unsigned c32 = (unsigned)c, cx = (unsigned)d.cx, cy = (unsigned)d.cy;
    const unsigned r = c32 / cx;
    x = (int)(c32 - r * cx);
    const unsigned q = r / cy;
    y = (int)(r - q * cy);
    z = (int)q;
}

// counts of one cell: triangles, vertices it creates
__device__ __forceinline__ void count_cell(const Tiling &t, unsigned own, int &nt, int &nv) {
    nt = t.nt;
    unsigned seen = 0;
    for (int i = 0; i < 3 * t.nt; ++i) seen |= 1u << t.row[i];
    nv = __popc(seen & own);
}

// ---------------------------------------------------------------- pass 1: classify
struct BlockSums {
    int nv, nt, na, pad;
};

__global__ __launch_bounds__(THREADS) void mc_classify_kernel(const float *__restrict__ vol, Dims d, double level,
                                                              unsigned *__restrict__ codes, BlockSums *__restrict__ block_counts,
                                                              float2 *__restrict__ block_minmax) {
    __shared__ int red[3][4];
    __shared__ float redf[2][4];
    const long long c0 = d.cell_begin + (long long)blockIdx.x * CELLS_PER_BLOCK;
    int nt_sum = 0, nv_sum = 0, na_sum = 0;
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (int r = 0; r < CELLS_PER_BLOCK / THREADS; ++r) {
        const long long c = c0 + r * THREADS + threadIdx.x;
        if (c >= d.cell_end) break;
        int x, y, z;
        cell_xyz(d, c, x, y, z);
        Cell cell;
        load_cell(vol, d, x, y, z, level, cell, lo, hi);
        unsigned code = 0;
        if (cell.index != 0 && cell.index != 255) {
            const Tiling t = select_tiling(cell.v, cell.index);
            int nt, nv;
            count_cell(t, owned_mask(x, y, z + d.zoff), nt, nv);
            nt_sum += nt;
            nv_sum += nv;
            na_sum += 1;
            code = encode_cell(t, nv);
        }
        if (codes) codes[c] = code;
    }
    // block reduce
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nt_sum += __shfl_xor(nt_sum, o);
        nv_sum += __shfl_xor(nv_sum, o);
        na_sum += __shfl_xor(na_sum, o);
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (lane == 0) { red[0][wave] = nv_sum; red[1][wave] = nt_sum; red[2][wave] = na_sum; redf[0][wave] = lo; redf[1][wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        BlockSums bs;
        bs.nv = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        bs.nt = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        bs.na = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        bs.pad = 0;
        block_counts[blockIdx.x] = bs;
        // the volume's min / max: per block here, reduced by the scan kernel (131 072 same-address atomics at 512^3 cost
        // 2.7 of this kernel's 3.0 ms)
        block_minmax[blockIdx.x] = make_float2(fminf(fminf(redf[0][0], redf[0][1]), fminf(redf[0][2], redf[0][3])),
                                               fmaxf(fmaxf(redf[1][0], redf[1][1]), fmaxf(redf[1][2], redf[1][3])));
    }
}

// ---------------------------------------------------------------- pass 2: exclusive scan of the block sums (one workgroup)
__global__ __launch_bounds__(1024) void mc_scan_kernel(const BlockSums *__restrict__ counts, BlockSums *__restrict__ offsets,
                                                       int nblocks, int *__restrict__ totals,
                                                       const float2 *__restrict__ block_minmax, float *__restrict__ minmax) {
    __shared__ int wsum[3][16];
    __shared__ int carry[3];
    __shared__ float wmm[2][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 3) carry[threadIdx.x] = 0;
    float lo = FLT_MAX, hi = -FLT_MAX;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        BlockSums v = {0, 0, 0, 0};
        if (i < nblocks) {
            v = counts[i];
            const float2 mm = block_minmax[i];
            lo = fminf(lo, mm.x);
            hi = fmaxf(hi, mm.y);
        }
        int a = v.nv, b = v.nt, c = v.na;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int na = __shfl_up(a, o), nb = __shfl_up(b, o), nc = __shfl_up(c, o);
            if (lane >= o) { a += na; b += nb; c += nc; }
        }
        if (lane == 63) { wsum[0][wave] = a; wsum[1][wave] = b; wsum[2][wave] = c; }
        __syncthreads();
        int pa = carry[0], pb = carry[1], pc = carry[2];
        for (int w = 0; w < wave; ++w) { pa += wsum[0][w]; pb += wsum[1][w]; pc += wsum[2][w]; }
        if (i < nblocks) {
            BlockSums o;
            o.nv = pa + a - v.nv; o.nt = pb + b - v.nt; o.na = pc + c - v.na; o.pad = 0;
            offsets[i] = o;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry[0] = pa + a; carry[1] = pb + b; carry[2] = pc + c; }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (lane == 0) { wmm[0][wave] = lo; wmm[1][wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[0] = carry[0]; totals[1] = carry[1]; totals[2] = carry[2];
        for (int w = 1; w < 16; ++w) { lo = fminf(lo, wmm[0][w]); hi = fmaxf(hi, wmm[1][w]); }
        minmax[0] = lo;
        minmax[1] = hi;
    }
}

// ---------------------------------------------------------------- pass 3: active-cell compaction with wavefront ballots
// One wave per 64 sweep-consecutive cells.  The rank of an active cell among the wave's active cells is the popcount of
// the ballot below its lane; the in-wave prefixes of the vertex count and triangle count (4 bits each) are
// sum_b 2^b * popc(ballot(bit b of the count) & lanes_below).  Waves of a block combine through LDS.
struct ActiveCell {
    unsigned cell, code;
    int vid0, tri0;
};

__global__ __launch_bounds__(THREADS) void mc_compact_kernel(const unsigned *__restrict__ codes, Dims d,
                                                             const BlockSums *__restrict__ block_offsets,
                                                             ActiveCell *__restrict__ alist) {
    __shared__ int wsum[3][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c0 = d.cell_begin + (long long)blockIdx.x * CELLS_PER_BLOCK;
    const BlockSums bo = block_offsets[blockIdx.x];
    int run_v = d.base_verts + bo.nv, run_t = d.base_faces + bo.nt, run_a = bo.na;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int r = 0; r < CELLS_PER_BLOCK / THREADS; ++r) {
        const long long c = c0 + r * THREADS + threadIdx.x;
        const unsigned code = (c < d.cell_end) ? codes[c] : 0u;
        const bool active = code != 0u;
        const unsigned long long am = __ballot(active);
        const int nt = code_nt(code), nv = code_nv(code);
        int pa = __popcll(am & below), pv = 0, pt = 0, wv = 0, wt = 0;
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) {
            const unsigned long long m = __ballot((nv >> bit) & 1);
            pv += __popcll(m & below) << bit;
            wv += __popcll(m) << bit;
        }
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) {
            const unsigned long long m = __ballot((nt >> bit) & 1);
            pt += __popcll(m & below) << bit;
            wt += __popcll(m) << bit;
        }
        __syncthreads();   // wsum of the previous round has been read by everyone
        if (lane == 0) { wsum[0][wave] = wv; wsum[1][wave] = wt; wsum[2][wave] = __popcll(am); }
        __syncthreads();
        int bv = 0, bt = 0, ba = 0, tv = 0, tt = 0, ta = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) { bv += wsum[0][w]; bt += wsum[1][w]; ba += wsum[2][w]; }
            tv += wsum[0][w]; tt += wsum[1][w]; ta += wsum[2][w];
        }
        if (active) {
            ActiveCell e;
            e.cell = (unsigned)c;
            e.code = code;
            e.vid0 = run_v + bv + pv;
            e.tri0 = run_t + bt + pt;
            alist[(size_t)(run_a + ba + pa)] = e;
        }
        run_v += tv; run_t += tt; run_a += ta;
    }
}

// lattice edge of cell edge e: which voxel it starts at and its axis (0 = x, 1 = y, 2 = z); 12 -> table 3
__device__ __forceinline__ void edge_slot(int e, int x, int y, int z, int &axis, int &vx, int &vy, int &vz) {
    // EDGE_D*[e][0..1] are the two corner offsets; the edge starts at the smaller one
    if (e == 12) { axis = 3; vx = x; vy = y; vz = z; return; }
    const int dx0 = MCL_EDGE_DX[2 * e], dx1 = MCL_EDGE_DX[2 * e + 1];
    const int dy0 = MCL_EDGE_DY[2 * e], dy1 = MCL_EDGE_DY[2 * e + 1];
    const int dz0 = MCL_EDGE_DZ[2 * e], dz1 = MCL_EDGE_DZ[2 * e + 1];
    axis = (dx0 != dx1) ? 0 : ((dy0 != dy1) ? 1 : 2);
    vx = x + min(dx0, dx1);
    vy = y + min(dy0, dy1);
    vz = z + min(dz0, dz1);
}

// ---------------------------------------------------------------- pass 4a: vertices, one thread per active cell
__global__ __launch_bounds__(THREADS) void mc_vertex_kernel(const float *__restrict__ vol, Dims d, double level,
                                                            const ActiveCell *__restrict__ alist, int nactive,
                                                            int *__restrict__ evid /* [4][nvox] */, float *__restrict__ verts,
                                                            float *__restrict__ normals, float *__restrict__ values,
                                                            int cap_verts) {
    const int a = blockIdx.x * THREADS + threadIdx.x;
    if (a >= nactive) return;
    const ActiveCell ac = alist[a];
    if (code_nv(ac.code) == 0) return;   // the cell creates no vertex
    const size_t nvox = (size_t)d.nx * d.ny * d.nz;
    int x, y, z;
    cell_xyz(d, (long long)ac.cell, x, y, z);
    Cell cell;
    float lo = 0, hi = 0;
    load_cell(vol, d, x, y, z, level, cell, lo, hi);
    const Tiling t = decode_cell(ac.code);
    const unsigned own = owned_mask(x, y, z + d.zoff);
    int vid = ac.vid0;
    unsigned seen = 0;
    for (int i = 0; i < 3 * t.nt; ++i) {
        const int e = t.row[i];
        if (seen & (1u << e)) continue;
        seen |= 1u << e;
        if (!(own & (1u << e))) continue;
        double px, py, pz;
        if (e == 12) {
            const int cx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, cy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, cz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
            double fx = 0, fy = 0, fz = 0, ff = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double w = 1.0 / (MC_EPS + fabs(cell.v[k]));
                fx += cx[k] * w; fy += cy[k] * w; fz += cz[k] * w; ff += w;
            }
            px = (double)x + fx / ff; py = (double)y + fy / ff; pz = (double)(z + d.zoff) + fz / ff;
        } else {
            const int dx1 = MCL_EDGE_DX[2 * e], dx2 = MCL_EDGE_DX[2 * e + 1];
            const int dy1 = MCL_EDGE_DY[2 * e], dy2 = MCL_EDGE_DY[2 * e + 1];
            const int dz1 = MCL_EDGE_DZ[2 * e], dz2 = MCL_EDGE_DZ[2 * e + 1];
            // corner number of (dx,dy,dz): v0..v7 = (0,0,0)(1,0,0)(1,1,0)(0,1,0)(0,0,1)(1,0,1)(1,1,1)(0,1,1)
            const int k1 = dz1 * 4 + (dy1 ? (dx1 ? 2 : 3) : (dx1 ? 1 : 0));
            const int k2 = dz2 * 4 + (dy2 ? (dx2 ? 2 : 3) : (dx2 ? 1 : 0));
            const double w1 = 1.0 / (MC_EPS + fabs(cell.v[k1])), w2 = 1.0 / (MC_EPS + fabs(cell.v[k2]));
            double fx = 0, fy = 0, fz = 0, ff = 0;
            fx += (double)dx1 * w1; fy += (double)dy1 * w1; fz += (double)dz1 * w1; ff += w1;
            fx += (double)dx2 * w2; fy += (double)dy2 * w2; fz += (double)dz2 * w2; ff += w2;
            px = (double)x + fx / ff; py = (double)y + fy / ff; pz = (double)(z + d.zoff) + fz / ff;
        }
        int axis, vx, vy, vz;
        edge_slot(e, x, y, z, axis, vx, vy, vz);
        evid[(size_t)axis * nvox + ((size_t)vz * d.ny + vy) * d.nx + vx] = vid;
        if (vid < cap_verts) {
            // output order (axis0, axis1, axis2) = (z, y, x)
            verts[3 * (size_t)vid + 0] = (float)pz;
            verts[3 * (size_t)vid + 1] = (float)py;
            verts[3 * (size_t)vid + 2] = (float)px;
            if (normals) { normals[3 * (size_t)vid] = 0.f; normals[3 * (size_t)vid + 1] = 0.f; normals[3 * (size_t)vid + 2] = 0.f; }
            if (values) values[vid] = 0.f;
        }
        ++vid;
    }
}

// ---------------------------------------------------------------- pass 4b: faces, values, normal accumulation
__global__ __launch_bounds__(THREADS) void mc_face_kernel(const float *__restrict__ vol, Dims d, double level,
                                                          const ActiveCell *__restrict__ alist, int nactive,
                                                          const int *__restrict__ evid, int *__restrict__ faces,
                                                          float *__restrict__ normals, float *__restrict__ values,
                                                          int cap_verts, int cap_faces) {
    const int a = blockIdx.x * THREADS + threadIdx.x;
    if (a >= nactive) return;
    const ActiveCell ac = alist[a];
    const size_t nvox = (size_t)d.nx * d.ny * d.nz;
    int x, y, z;
    cell_xyz(d, (long long)ac.cell, x, y, z);
    const Tiling t = decode_cell(ac.code);
    const int tri0 = ac.tri0;
    Cell cell;
    for (int k = 0; k < 8; ++k) cell.v[k] = 0.0;
    if (normals || values) {   // the corner values are only needed for these
        float lo = 0, hi = 0;
        load_cell(vol, d, x, y, z, level, cell, lo, hi);
    }
    const double *v = cell.v;
    // per-cell quantities for values / normals (see oracle/mc_oracle.c for the black-box verified quirks)
    double vlo = 0.0, vhi = 0.0;
    for (int k = 0; k < 8; ++k) { if (v[k] > vhi) vhi = v[k]; if (v[k] < vlo) vlo = v[k]; }
    const float vrange = (float)(vhi - vlo);
    const double g[8][3] = {
        {v[0] - v[1], v[0] - v[3], v[0] - v[4]}, {v[0] - v[1], v[1] - v[2], v[1] - v[5]},
        {v[3] - v[2], v[1] - v[2], v[2] - v[6]}, {v[3] - v[2], v[0] - v[3], v[3] - v[7]},
        {v[4] - v[5], v[4] - v[7], v[0] - v[4]}, {v[4] - v[5], v[5] - v[6], v[1] - v[5]},
        {v[7] - v[6], v[5] - v[6], v[2] - v[6]}, {v[7] - v[6], v[4] - v[7], v[3] - v[7]}};
    double c12g[3] = {0, 0, 0};
    if (normals) {
        double gy = 0, gz = 0;
        for (int k = 0; k < 8; ++k) {
            const double w = 1.0 / (MC_EPS + fabs(v[k]));
            gy += w * g[k][1];
            gz += w * g[k][2];
        }
        c12g[0] = gz; c12g[1] = gy; c12g[2] = 0.0;  // quirk of the compiled core: (Gz, Gy, 0)
    }
    for (int i = 0; i < t.nt; ++i) {
        int vid[3];
        for (int j = 0; j < 3; ++j) {
            const int e = t.row[3 * i + j];
            int axis, vx, vy, vz;
            edge_slot(e, x, y, z, axis, vx, vy, vz);
            vid[j] = evid[(size_t)axis * nvox + ((size_t)vz * d.ny + vy) * d.nx + vx];
            if (vid[j] < cap_verts) {
                if (values) atomicMax(reinterpret_cast<int *>(values) + vid[j], __float_as_int(vrange));
                if (normals) {
                    float *n = normals + 3 * (size_t)vid[j];
                    if (e == 12) {
                        atomicAdd(n + 0, (float)c12g[0]); atomicAdd(n + 1, (float)c12g[1]); atomicAdd(n + 2, (float)c12g[2]);
                    } else {
                        const int dx1 = MCL_EDGE_DX[2 * e], dx2 = MCL_EDGE_DX[2 * e + 1];
                        const int dy1 = MCL_EDGE_DY[2 * e], dy2 = MCL_EDGE_DY[2 * e + 1];
                        const int dz1 = MCL_EDGE_DZ[2 * e], dz2 = MCL_EDGE_DZ[2 * e + 1];
                        const int i1 = dz1 * 4 + dy1 * 2 + dx1, i2 = dz2 * 4 + dy2 * 2 + dx2;  // xyz-bit index
                        const int k1 = dz1 * 4 + (dy1 ? (dx1 ? 2 : 3) : (dx1 ? 1 : 0));
                        const int k2 = dz2 * 4 + (dy2 ? (dx2 ? 2 : 3) : (dx2 ? 1 : 0));
                        const double w1 = 1.0 / (MC_EPS + fabs(v[k1])), w2 = 1.0 / (MC_EPS + fabs(v[k2]));
                        // the core indexes its corner-number gradient table with the xyz-bit index (quirk)
                        for (int ax = 0; ax < 3; ++ax) {
                            atomicAdd(n + ax, (float)(g[i1][ax] * w1));
                            atomicAdd(n + ax, (float)(g[i2][ax] * w2));
                        }
                    }
                }
            }
        }
        const int tri = tri0 + i;
        if (tri < cap_faces) {
            // rows reversed (gradient_direction='descent')
            faces[3 * (size_t)tri + 0] = vid[2];
            faces[3 * (size_t)tri + 1] = vid[1];
            faces[3 * (size_t)tri + 2] = vid[0];
        }
    }
}

// ---------------------------------------------------------------- pass 5: normalise and flip normals to (axis0, axis1, axis2)
__global__ void mc_normalize_kernel(float *__restrict__ normals, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float *p = normals + 3 * (size_t)i;
    const float nx = p[0], ny = p[1], nz = p[2];
    const double len = sqrt((double)nx * nx + (double)ny * ny + (double)nz * nz);
    float ox = nx, oy = ny, oz = nz;
    if (len > 0.0) { ox = (float)(nx / len); oy = (float)(ny / len); oz = (float)(nz / len); }
    p[0] = oz; p[1] = oy; p[2] = ox;
}


// verts_world = mat[:3,:3] @ v + mat[:3,3]  in float64 (lib/mesh_util.py:42-43,47-48)
struct Affine { double m[12]; };
__global__ void transform_points_kernel(const float *__restrict__ v, int n, Affine a, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = v[3 * (size_t)i], y = v[3 * (size_t)i + 1], z = v[3 * (size_t)i + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) out[3 * (size_t)i + r] = ((a.m[4 * r] * x + a.m[4 * r + 1] * y) + a.m[4 * r + 2] * z) + a.m[4 * r + 3];
}

// ---------------------------------------------------------------- slab mode (one volume split along axis 0 over ranks)
// The cells of a slab's first layer reference the vertices of the x- and y-edges in its plane 0, which the slab below
// created and numbered.  Until those ids arrive the tables hold a reference to the edge itself: -(2 + slot),
// slot = axis * ny * nx + y * nx + x; mc_slab_fixup_kernel resolves them once the ranks have exchanged ids and counts.
__global__ void mc_slab_refs_kernel(int *__restrict__ evid, size_t nvox, int plane) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * plane) return;
    const int axis = i / plane, r = i - axis * plane;
    evid[(size_t)axis * nvox + r] = -(2 + i);
}

__global__ void mc_slab_fixup_kernel(int *__restrict__ faces, long long n, int own_offset, const int *__restrict__ below_ids,
                                     int below_offset) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = faces[i];
    faces[i] = v >= 0 ? v + own_offset : below_ids[-v - 2] + below_offset;
}

}  // namespace mc
}  // namespace surs

using namespace surs;
using namespace surs::mc;

static int mc_nblocks(long long ncells) { return (int)((ncells + CELLS_PER_BLOCK - 1) / CELLS_PER_BLOCK); }

static size_t mc_ws_layout(int n0, int n1, int n2, size_t off[5]) {
    const long long ncells = (long long)(n0 - 1) * (n1 - 1) * (n2 - 1);
    const size_t nb = (size_t)mc_nblocks(ncells);
    const size_t nvox = (size_t)n0 * n1 * n2;
    size_t o = 0;
    off[0] = o; o += align_up(nb * sizeof(BlockSums), 256);         // block sums
    off[1] = o; o += align_up(nb * sizeof(BlockSums), 256);         // block offsets
    off[2] = o; o += 256 + align_up(nb * sizeof(float2), 256);      // min/max, totals; per-block min/max
    off[3] = o; o += align_up((size_t)ncells * sizeof(unsigned), 256);   // cell codes
    off[4] = o; o += align_up(4 * nvox * sizeof(int), 256);         // edge -> vertex id tables
    return o;   // the active-cell list follows; its size is known after pass 2 (worst case: every cell)
}

extern "C" size_t surs_mc_workspace_bytes(int n0, int n1, int n2) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return 0;
    size_t off[5];
    const long long ncells = (long long)(n0 - 1) * (n1 - 1) * (n2 - 1);
    return mc_ws_layout(n0, n1, n2, off) + align_up((size_t)ncells * sizeof(ActiveCell), 256);
}

// One contiguous range of cells in sweep order, [cell_begin, cell_end): classify, scan, and - unless count_only -
// compact, vertices, faces; vertex / face ids continue from run->n_verts / run->n_faces, which are advanced (also on
// SURS_E_CAPACITY, so that the caller knows the sizes), run->vmin / vmax are widened.  Synchronises once (the host
// needs the counts to size the launches that follow).
static int mc_range(const float *vol, int n0, int n1, int n2, long long cell_begin, long long cell_end, double level,
                    void *workspace, size_t workspace_bytes, float *verts, float *normals, float *values, int cap_verts,
                    int32_t *faces, int cap_faces, bool count_only, surs_mc_counts *run, hipStream_t st, int zoff = 0) {
    Dims d;
    d.zoff = zoff;
    d.nz = n0; d.ny = n1; d.nx = n2;
    d.cz = n0 - 1; d.cy = n1 - 1; d.cx = n2 - 1;
    d.ncells = (long long)d.cz * d.cy * d.cx;
    SURS_REQUIRE(d.ncells < (1ll << 32), "volume too large");
    SURS_REQUIRE(cell_begin >= 0 && cell_begin <= cell_end && cell_end <= d.ncells, "bad cell range");
    if (cell_begin == cell_end) return 0;
    d.cell_begin = cell_begin;
    d.cell_end = cell_end;
    d.base_verts = run->n_verts;
    d.base_faces = run->n_faces;
    const int nb = mc_nblocks(cell_end - cell_begin);
    size_t off[5];
    const size_t fixed = mc_ws_layout(n0, n1, n2, off);
    char *ws = (char *)workspace;
    BlockSums *bcounts = (BlockSums *)(ws + off[0]);
    BlockSums *boffs = (BlockSums *)(ws + off[1]);
    float *minmax = (float *)(ws + off[2]);
    int *totals = (int *)(ws + off[2] + 16);
    float2 *bminmax = (float2 *)(ws + off[2] + 256);
    unsigned *codes = (unsigned *)(ws + off[3]);
    int *evid = (int *)(ws + off[4]);
    ActiveCell *alist = (ActiveCell *)(ws + fixed);

    hipLaunchKernelGGL(mc_classify_kernel, dim3(nb), dim3(THREADS), 0, st, vol, d, level, count_only ? (unsigned *)nullptr : codes,
                       bcounts, bminmax);
    SURS_LAUNCH_CHECK();
    hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, st, bcounts, boffs, nb, totals, bminmax, minmax);
    SURS_LAUNCH_CHECK();
    struct { float mm[2]; unsigned pad[2]; int tot[3]; } host;
    SURS_HIP_CHECK(hipMemcpyAsync(&host, minmax, sizeof(host), hipMemcpyDeviceToHost, st));
    SURS_HIP_CHECK(hipStreamSynchronize(st));
    run->vmin = fminf(run->vmin, host.mm[0]);
    run->vmax = fmaxf(run->vmax, host.mm[1]);
    const int nactive = host.tot[2];
    run->n_verts = d.base_verts + host.tot[0];
    run->n_faces = d.base_faces + host.tot[1];
    if (count_only || nactive == 0) return 0;
    if (run->n_verts > cap_verts || run->n_faces > cap_faces)
        return fail(SURS_E_CAPACITY, "output capacity too small: need %d vertices, %d faces", run->n_verts, run->n_faces);
    hipLaunchKernelGGL(mc_compact_kernel, dim3(nb), dim3(THREADS), 0, st, codes, d, boffs, alist);
    SURS_LAUNCH_CHECK();
    const int ab = ceil_div(nactive, THREADS);
    hipLaunchKernelGGL(mc_vertex_kernel, dim3(ab), dim3(THREADS), 0, st, vol, d, level, alist, nactive, evid, verts, normals,
                       values, cap_verts);
    SURS_LAUNCH_CHECK();
    hipLaunchKernelGGL(mc_face_kernel, dim3(ab), dim3(THREADS), 0, st, vol, d, level, alist, nactive, evid, faces, normals, values,
                       cap_verts, cap_faces);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_mc_lewiner(const float *vol, int n0, int n1, int n2, double level, void *workspace,
                               size_t workspace_bytes, float *verts, float *normals, float *values, int cap_verts,
                               int32_t *faces, int cap_faces, surs_mc_counts *counts, void *stream) {
    SURS_REQUIRE(vol && workspace && counts, "null argument");
    SURS_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "Input array must be at least 2x2x2.");
    SURS_REQUIRE(workspace_bytes >= surs_mc_workspace_bytes(n0, n1, n2), "workspace too small");
    SURS_REQUIRE((long long)n0 * n1 * n2 < (1ll << 31) * 4, "volume too large");
    hipStream_t st = as_stream(stream);
    const long long ncells = (long long)(n0 - 1) * (n1 - 1) * (n2 - 1);
    const bool count_only = !verts || !faces;
    counts->n_verts = 0;
    counts->n_faces = 0;
    counts->vmin = FLT_MAX;
    counts->vmax = -FLT_MAX;
    // the level-range and no-surface errors come before the capacity error, as in one pass over the whole volume
    surs_mc_counts probe = *counts;
    int rc = mc_range(vol, n0, n1, n2, 0, ncells, level, workspace, workspace_bytes, verts, normals, values, cap_verts, faces,
                      cap_faces, count_only, &probe, st);
    *counts = probe;
    if (rc && rc != SURS_E_CAPACITY) return rc;
    if (level < (double)counts->vmin || level > (double)counts->vmax)
        return fail(SURS_E_LEVEL_RANGE, "Surface level must be within volume data range.");
    if (counts->n_verts == 0) return fail(SURS_E_NO_SURFACE, "No surface found at the given iso value.");
    if (rc) return rc;
    if (count_only) return 0;
    if (normals) {
        hipLaunchKernelGGL(mc_normalize_kernel, dim3(ceil_div(counts->n_verts, 256)), dim3(256), 0, st, normals, counts->n_verts);
        SURS_LAUNCH_CHECK();
    }
    SURS_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int surs_mc_lewiner_range(const float *vol, int n0, int n1, int n2, int layer_begin, int layer_end, double level,
                                     void *workspace, size_t workspace_bytes, float *verts, float *normals, float *values,
                                     int cap_verts, int32_t *faces, int cap_faces, surs_mc_counts *run, void *stream) {
    SURS_REQUIRE(vol && workspace && run && verts && faces, "null argument");
    SURS_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "Input array must be at least 2x2x2.");
    SURS_REQUIRE(workspace_bytes >= surs_mc_workspace_bytes(n0, n1, n2), "workspace too small");
    SURS_REQUIRE(layer_begin >= 0 && layer_begin <= layer_end && layer_end <= n0 - 1, "bad layer range");
    const long long per_layer = (long long)(n1 - 1) * (n2 - 1);
    return mc_range(vol, n0, n1, n2, layer_begin * per_layer, layer_end * per_layer, level, workspace, workspace_bytes, verts,
                    normals, values, cap_verts, faces, cap_faces, false, run, as_stream(stream));
}

extern "C" int surs_mc_lewiner_range_slab(const float *vol, int n0, int n1, int n2, int layer_begin, int layer_end, double level,
                                          void *workspace, size_t workspace_bytes, float *verts, int cap_verts, int32_t *faces,
                                          int cap_faces, surs_mc_counts *run, int z_offset, void *stream) {
    SURS_REQUIRE(vol && workspace && run && verts && faces, "null argument");
    SURS_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "Input array must be at least 2x2x2.");
    SURS_REQUIRE(workspace_bytes >= surs_mc_workspace_bytes(n0, n1, n2), "workspace too small");
    SURS_REQUIRE(layer_begin >= 0 && layer_begin <= layer_end && layer_end <= n0 - 1, "bad layer range");
    SURS_REQUIRE(z_offset >= 0, "negative z_offset");
    hipStream_t st = as_stream(stream);
    if (z_offset > 0 && layer_begin == 0) {
        size_t off[5];
        mc_ws_layout(n0, n1, n2, off);
        int *evid = (int *)((char *)workspace + off[4]);
        const int plane = n1 * n2;
        hipLaunchKernelGGL(mc_slab_refs_kernel, dim3(ceil_div(2 * plane, 256)), dim3(256), 0, st, evid, (size_t)n0 * n1 * n2, plane);
        SURS_LAUNCH_CHECK();
    }
    const long long per_layer = (long long)(n1 - 1) * (n2 - 1);
    return mc_range(vol, n0, n1, n2, layer_begin * per_layer, layer_end * per_layer, level, workspace, workspace_bytes, verts,
                    nullptr, nullptr, cap_verts, faces, cap_faces, false, run, st, z_offset);
}

extern "C" int surs_mc_slab_top_ids(const void *workspace, size_t workspace_bytes, int n0, int n1, int n2, int32_t *ids, void *stream) {
    SURS_REQUIRE(workspace && ids && n0 >= 2 && n1 >= 2 && n2 >= 2, "bad argument");
    SURS_REQUIRE(workspace_bytes >= surs_mc_workspace_bytes(n0, n1, n2), "workspace too small");
    size_t off[5];
    mc_ws_layout(n0, n1, n2, off);
    const int *evid = (const int *)((const char *)workspace + off[4]);
    const size_t nvox = (size_t)n0 * n1 * n2, plane = (size_t)n1 * n2;
    for (int axis = 0; axis < 2; ++axis)
        SURS_HIP_CHECK(hipMemcpyAsync(ids + axis * plane, evid + axis * nvox + (size_t)(n0 - 1) * plane, plane * sizeof(int),
                                      hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}

extern "C" int surs_mc_slab_fixup(int32_t *faces, long long n_faces, int own_offset, const int32_t *below_ids, int below_offset,
                                  void *stream) {
    if (n_faces <= 0) return 0;
    SURS_REQUIRE(faces, "null argument");
    const long long n = 3 * n_faces;
    hipLaunchKernelGGL(mc_slab_fixup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), faces, n, own_offset,
                       below_ids, below_offset);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_mc_normalize(float *normals, int n_verts, void *stream) {
    if (n_verts <= 0) return 0;
    SURS_REQUIRE(normals, "null argument");
    hipLaunchKernelGGL(mc_normalize_kernel, dim3(ceil_div(n_verts, 256)), dim3(256), 0, as_stream(stream), normals, n_verts);
    SURS_LAUNCH_CHECK();
    return 0;
}

extern "C" int surs_transform_points(const float *verts, int n, const double *mat, double *out, void *stream) {
    if (n == 0) return 0;
    SURS_REQUIRE(verts && mat && out && n > 0, "bad argument");
    Affine a;
    for (int i = 0; i < 12; ++i) a.m[i] = mat[i];
    hipLaunchKernelGGL(transform_points_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), verts, n, a, out);
    SURS_LAUNCH_CHECK();
    return 0;
}
