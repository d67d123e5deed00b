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
// So:  pass 1  count: every thread streams the corner rows of four consecutive cells (aligned 16-byte loads), decides
//              inside / outside per voxel with one float compare and lists the cells the surface crosses in LDS (wavefront
//              ballots for the ranks); the listed cells are classified (MC33 tests in double; a 256-entry table for the
//              cube indices that need no test) and vertices / triangles / active cells are summed per block of 1024
//              sweep-consecutive cells, with the volume's min / max.  Nothing is written per cell.
//      pass 2  exclusive scan of the block sums, two levels (1024 blocks per group, then the groups)
//      pass 3  emit: the same classification again, only in blocks that have active cells; one 16-byte entry per active
//              cell in sweep order: (cell, code = table / offset / counts, first vertex id, first triangle id)
//      pass 4a vertices, one thread per ACTIVE cell: positions; ids stored in 4 per-voxel tables (x-edge, y-edge, z-edge
//              starting at the voxel, centre of the cell whose corner 0 it is) that hold a RING of planes (option mc_ring): the sweep is
//              walked in chunks of ring - 1 cell layers (mc_chunked), a layer only touches the ids of its two planes
//      pass 4b faces (+ values by atomic max, normals by atomic add): vertex ids looked up in the tables
//      pass 5  normalise normals.
// HBM-bound by design: the volume is read once (pass 1) plus the blocks with a surface once more (pass 3: a few per cent
// for a body-sized surface), then the active cells only.  Measured at 512^3 on a body-sized blob (4.3 x 10^5 vertices):
// 0.67 ms of kernels (count 0.37, emit 0.19, vertices + faces 0.09), pass 1 being bound by its ~100 integer / compare
// instructions per cell, not by the 537 MB it reads.  All arithmetic that decides topology or positions is double, written
// exactly as the Cython core evaluates it (compiled with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <mutex>

#include "surs_common.h"

namespace surs {
namespace mc {

#define MCL_QUAL __device__ const
#include "mc_luts.inc"
#undef MCL_QUAL

// the Cython core defines FLT_EPSILON = np.spacing(1.0): DOUBLE epsilon (black-box verified, see oracle)
#define MC_EPS 2.220446049250313e-16

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int CELLS_PER_BLOCK = 1024;
constexpr int THREADS = 256;

struct Dims {
    int nz, ny, nx;     // volume dims (axis 0, 1, 2)
    int cz, cy, cx;     // cells per axis
    long long ncells;
    long long cell_begin, cell_end;   // the flat (sweep-order) cell range this call processes (whole layers)
    int prow;                         // padded row length of the classification passes: cx rounded up to a multiple of 4
    long long q_begin, q_end;         // the same range as padded indices q = (z * cy + y) * prow + x
    int base_verts, base_faces;      // vertices / triangles produced by earlier ranges
    int zoff;                        // slab mode: index of the volume's plane 0 in the whole grid (0 = the grid's bottom)
    int ring;                        // planes of the edge -> vertex-id tables: plane z lives in slot z % ring (mc_ring)
    // q / prow and row / cy by multiplication (fast_div): n / d = mulhi(n, m) >> s for every n < 2^31, with m = ceil(2^(32 + s) / d),
    // s = ceil(log2 d) - 1 (error of m below d, times n below 2^(32 + s)); fastdiv = 0 (d = 1 or q_end >= 2^31): plain division
    unsigned m_prow, s_prow, m_cy, s_cy;
    int fastdiv;
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

// ---------------------------------------------------------------- passes 1 and 3: classify (count) and classify again (emit)
// A block = 1024 sweep-consecutive cells, a thread = 8 consecutive cells (two quads).  Inside one grid row (the common case) the four
// cells share their corner rows: 4 rows x 5 consecutive floats instead of 4 x 8 scattered loads, each voxel compared with
// the level once.  Pass 1 (count) only sums vertices / triangles / active cells and min / max per block: nothing is written
// per cell.  After the scan, pass 3 (emit) runs the same classification again - but only in blocks that have active cells
// (for a body-sized surface a few per cent of the blocks: the volume is read once plus that) - and writes the active cells
// in sweep order with their first vertex / triangle numbers.  The cell code (table, offset, counts) lives in registers.
struct BlockSums {
    int nv, nt, na, pad;
};

struct ActiveCell {
    unsigned cell, code;
    int vid0, tri0;
};

// the classification passes (count, re-classifying emit): two waves per 1024-cell block, eight cells = two quads per thread
constexpr int SCAN_THREADS = 128, SCAN_WAVES = SCAN_THREADS / 64;
constexpr int CELLS_PER_THREAD = CELLS_PER_BLOCK / SCAN_THREADS;
static_assert(CELLS_PER_THREAD % 4 == 0, "mc_scan_block handles whole quads of cells per thread");

// Cube indices whose MC33 case needs no face / interior test (cases 1, 2, 5, 8, 9, 11, 14: every cell of a smooth surface)
// resolve through a 256-entry table filled once per device: the cell code without its vertex count, and the set of edges
// its triangles use (the vertex count is the popcount of that set masked with the edges the cell owns).  The general path -
// select_tiling and a walk over the triangle list, a chain of dependent look-ups - remains for the ambiguous cases.
__device__ unsigned g_fast_code[256];
__device__ unsigned g_fast_seen[256];

__global__ void mc_fast_init_kernel() {
    const int index = threadIdx.x;
    const int mccase = MCL_CASES[2 * index];
    unsigned code = 0, seen = 0;
    if (mccase == 1 || mccase == 2 || mccase == 5 || mccase == 8 || mccase == 9 || mccase == 11 || mccase == 14) {
        const double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // not read for these cases
        const Tiling t = select_tiling(v, index);
        for (int i = 0; i < 3 * t.nt; ++i) seen |= 1u << t.row[i];
        code = encode_cell(t, 0);
    }
    g_fast_code[index] = code;
    g_fast_seen[index] = seen;
}

__device__ __forceinline__ unsigned classify_cell(const double *v, int index, unsigned own) {
    const unsigned fast = g_fast_code[index];
    if (fast != 0u) return fast | ((unsigned)__popc(g_fast_seen[index] & own) << 4);
    const Tiling t = select_tiling(v, index);
    int nt, nv;
    count_cell(t, own, nt, nv);
    return encode_cell(t, nv);
}

// Threads walk a PADDED cell index q = row * prow + x (row = z * cy + y, prow = cx rounded up to a multiple of 4; cells with
// x >= cx do not exist): monotone in the sweep order, a thread's cells come in quads q .. q+3 that lie in one row and start at
// x % 4 == 0, so with nx % 4 == 0 each corner row of a quad is one aligned 16-byte load.
//
// Both passes run in two phases.  Phase 1 streams: every thread reads the corner rows of its two quads, inside / outside is
// decided per voxel with one float compare (levelf = the largest float <= level: for a float f, f > levelf is exactly
// (double)f - level > 0), and the cells the surface crosses are appended, in sweep order, to a list in LDS with their cube
// indices (wavefront ballots for the ranks).  Phase 2 works on that list with all lanes busy: one entry per thread and round,
// classified from the cube index alone or - the ambiguous MC33 cases - after re-reading the corners (tests in double).  A block
// without surface - nearly every block of a body-sized field - ends after phase 1.
struct CellList {
    unsigned short x[CELLS_PER_BLOCK];     // position of the cell in the block (padded index - block start)
    unsigned char index[CELLS_PER_BLOCK];  // its cube index
    int wave_count[SCAN_WAVES];
    float wave_lo[SCAN_WAVES], wave_hi[SCAN_WAVES];   // the count pass: the waves' voxel minima / maxima
    int n;
};

__device__ __forceinline__ void decode_q(const Dims &d, long long q, int &x, int &y, int &z) {
    const unsigned q32 = (unsigned)q, pr = (unsigned)d.prow, cy = (unsigned)d.cy;
    unsigned row, zz;
    if (d.fastdiv) {   // uniform
        row = __umulhi(q32, d.m_prow) >> d.s_prow;
        zz = __umulhi(row, d.m_cy) >> d.s_cy;
    } else {
        row = q32 / pr;
        zz = row / cy;
    }
    x = (int)(q32 - row * pr);
    y = (int)(row - zz * cy);
    z = (int)zz;
}

// wavefront reductions on the vector ALU's data-parallel primitives (row shifts inside the rows of 16 lanes, then the row
// broadcasts): six dependent vector instructions, result in lane 63 - the shuffle form goes through the LDS crossbar once per step
// (v_min_f32 / v_max_f32 with the DPP modifier on the first source: one instruction per step; lanes without a source lane - bound_ctrl
//  off - keep their value.  Written through the builtins every step was a move, two canonicalising maxima and the minimum.)
__device__ __forceinline__ float wave_min_to_lane63(float x) {
    asm("s_nop 1\n\t"   // (a VGPR written by the instruction before must not be read through DPP for two wait states)
        "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"   // lane 15 of every row holds the row's minimum
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"   // into rows 1 and 3
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"         // into rows 2 and 3
        : "+v"(x));
    return x;
}
__device__ __forceinline__ float wave_max_to_lane63(float x) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(x));
    return x;
}

// phase 1 for the block starting at padded index q0: fills `list` (sweep order) and widens lo / hi by the block's voxels
// Returns the number of listed cells (uniform).  FIRST: the workgroup has not used `list` before (no barrier in front of it);
// MINMAX: each wave leaves its voxel minimum / maximum in list.wave_lo / wave_hi.  A block without listed cells - nearly all of a
// body-sized field - returns after ONE barrier.
template <bool FIRST, bool MINMAX>
__device__ __forceinline__ int mc_scan_block(const float *__restrict__ vol, const Dims &d, float levelf, long long q0,
                                             CellList &list, float &lo, float &hi, int *__restrict__ nan_flag = nullptr) {
    constexpr int NQ = CELLS_PER_THREAD / 4;   // quads of four cells per thread, consecutive in the padded index
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long q = q0 + (long long)threadIdx.x * CELLS_PER_THREAD;
    unsigned idx[CELLS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CELLS_PER_THREAD; ++i) idx[i] = 0;
    bool wave_empty = false;   // (uniform) known without looking at the lanes' cube indices: the rank arithmetic is skipped
    if ((d.nx & 3) == 0) {
        // Rows of whole 16-byte quads (prow == nx): the inside / outside decisions are made per VOXEL, as wavefront masks - a
        // vector compare writes the 64 lanes' results into a scalar register pair, so bit L of m[h][k][j] is voxel j of lane L's
        // quad h in row k - and a cell's activity (its 8 corners not all on one side) is a handful of 64-bit scalar AND / ORs per
        // quad position instead of ~ 30 vector instructions per cell.  The voxel behind a lane's first quad is its second quad's
        // first, the one behind its last quad the next lane's first (the mask shifted by one lane; only lane 63 reads it).  A
        // wavefront without an active cell - nearly all of a body-sized field - is done here; otherwise the cube indices are
        // assembled per lane.  (Two quads per lane: the per-wave work - decode, reductions, ranks, barrier - is spread over 512
        // cells; the pass is bound by instruction issue, vector and scalar together)
        const size_t sy = (size_t)d.nx, sz = (size_t)d.nx * d.ny;
        bool valid[NQ];
        int y[NQ], z[NQ], ncell[NQ];
        f32x4_t v[NQ][4];
        const float *plast = vol;
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            const long long qh = q + 4 * h;
            valid[h] = qh < d.q_end;
            int x = 0;
            y[h] = z[h] = 0;
            if (valid[h]) decode_q(d, qh, x, y[h], z[h]);
            const float *p = vol + (valid[h] ? (size_t)z[h] * sz + (size_t)y[h] * sy + x : (size_t)0);   // past the range: voxels 0..3 again
            ncell[h] = valid[h] ? max(0, min(4, d.cx - x)) : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[h][k] = *reinterpret_cast<const f32x4_t *>(p + (k & 1) * sy + (k >> 1) * sz);
            plast = p;
        }
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane == 63 && ncell[NQ - 1] == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = plast[(k & 1) * sy + (k >> 1) * sz + 4];
        }
        // min / max / NaN: every voxel is in row 0 of some quad, except the last plane of axis 1 (row 1 at the last cell row) and
        // of axis 0 (rows 2, 3 in the last cell layer of the volume).  An unordered compare of two values is true if either is a NaN.
        unsigned long long nanm = 0ull;
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            if (valid[h]) {
                lo = fminf(lo, fminf(fminf(v[h][0][0], v[h][0][1]), fminf(v[h][0][2], v[h][0][3])));
                hi = fmaxf(hi, fmaxf(fmaxf(v[h][0][0], v[h][0][1]), fmaxf(v[h][0][2], v[h][0][3])));
            }
            nanm |= (__ballot(__builtin_isunordered(v[h][0][0], v[h][0][1])) | __ballot(__builtin_isunordered(v[h][0][2], v[h][0][3]))) &
                    __ballot(valid[h]);
            const bool ylast = valid[h] && y[h] == d.cy - 1, zlast = valid[h] && z[h] == d.cz - 1;
            if (__ballot(ylast || zlast) != 0ull) {
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const bool use = (k == 1 && ylast) || (k == 2 && zlast) || (k == 3 && ylast && zlast);
                    if (use) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            lo = fminf(lo, v[h][k][j]);
                            hi = fmaxf(hi, v[h][k][j]);
                        }
                    }
                    nanm |= __ballot(use && (v[h][k][0] != v[h][k][0] || v[h][k][1] != v[h][k][1] || v[h][k][2] != v[h][k][2] ||
                                             v[h][k][3] != v[h][k][3]));
                }
            }
        }
        // fminf / fmaxf skip NaNs: report them (an overflowed f16 activation of the fp32-grade sweep shows up so)
        if (nanm != 0ull && nan_flag && lane == 0) atomicOr(nan_flag, 1);
        // voxel COLUMNS (the four rows of one quad position): all inside / some inside; column 4 NQ = the voxel behind the last quad
        unsigned long long nxt[4];        // its four rows, per lane (bit L: the voxel behind lane L's last quad)
        unsigned long long call[4 * NQ + 1], csome[4 * NQ + 1];
#pragma unroll
        for (int h = 0; h < NQ; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long m0 = __ballot(v[h][0][j] > levelf), m1 = __ballot(v[h][1][j] > levelf),
                                         m2 = __ballot(v[h][2][j] > levelf), m3 = __ballot(v[h][3][j] > levelf);
                call[4 * h + j] = (m0 & m1) & (m2 & m3);
                csome[4 * h + j] = (m0 | m1) | (m2 | m3);
                if (h == 0 && j == 0) { nxt[0] = m0 >> 1; nxt[1] = m1 >> 1; nxt[2] = m2 >> 1; nxt[3] = m3 >> 1; }
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) nxt[k] |= __ballot(e[k] > levelf) & (1ull << 63);
        call[4 * NQ] = (nxt[0] & nxt[1]) & (nxt[2] & nxt[3]);
        csome[4 * NQ] = (nxt[0] | nxt[1]) | (nxt[2] | nxt[3]);
        // cell i of quad h = columns 4 h + i and 4 h + i + 1 (a quad whose fourth cell exists lies in one row with its successor)
        unsigned long long any = 0ull;
#pragma unroll
        for (int h = 0; h < NQ; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                any |= (csome[4 * h + i] | csome[4 * h + i + 1]) & ~(call[4 * h + i] & call[4 * h + i + 1]) & __ballot(i < ncell[h]);
        wave_empty = any == 0ull;
        if (any != 0ull) {
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                unsigned in[4];   // bit j of in[k]: voxel j of row k (of quad h and the voxel behind it) is inside
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    in[k] = (h + 1 < NQ) ? ((v[h + 1 < NQ ? h + 1 : h][k][0] > levelf ? 1u : 0u) << 4)
                                         : ((unsigned)((nxt[k] >> lane) & 1ull) << 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) in[k] |= (v[h][k][j] > levelf ? 1u : 0u) << j;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned a0 = in[0] >> i, a1 = in[1] >> i, a2 = in[2] >> i, a3 = in[3] >> i;
                    unsigned ci = (a0 & 1) | (a0 & 2) | ((a1 & 2) << 1) | ((a1 & 1) << 3) | ((a2 & 1) << 4) | ((a2 & 2) << 4) |
                                  ((a3 & 2) << 5) | ((a3 & 1) << 7);
                    if (i >= ncell[h]) ci = 0;
                    idx[4 * h + i] = ci;
                }
            }
        }
    } else {
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            const long long qh = q + 4 * h;
            if (qh >= d.q_end) continue;
            int x, y, z;
            decode_q(d, qh, x, y, z);
            const int ncell = max(0, min(4, d.cx - x));
            if (ncell <= 0) continue;
            const size_t sy = (size_t)d.nx, sz = (size_t)d.nx * d.ny;
            const float *p = vol + (size_t)z * sz + (size_t)y * sy + x;
            float r[4][5];   // rows (y, z), (y+1, z), (y, z+1), (y+1, z+1)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 5; ++j) r[k][j] = p[(k & 1) * sy + (k >> 1) * sz + min(j, ncell)];
            // min / max: every voxel is covered by row 0 of some quad, except the last plane of axis 1 (row 1 when y is the
            // last cell row) and of axis 0 (rows 2, 3 when z is the last cell layer of the volume)
            const bool ylast = y == d.cy - 1, zlast = z == d.cz - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool use = k == 0 || (k == 1 && ylast) || (k == 2 && zlast) || (k == 3 && ylast && zlast);
                if (use) {
                    bool bad = false;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        lo = fminf(lo, r[k][j]);
                        hi = fmaxf(hi, r[k][j]);
                        bad |= r[k][j] != r[k][j];
                    }
                    // fminf / fmaxf skip NaNs: report them (an overflowed f16 activation of the fp32-grade sweep shows up so)
                    if (bad && nan_flag) atomicOr(nan_flag, 1);
                }
            }
            unsigned in[4];   // bit j of in[k]: voxel j of row k is inside
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                in[k] = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) in[k] |= (r[k][j] > levelf ? 1u : 0u) << j;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // corners v0..v7 = (x,y,z) (x+1,y,z) (x+1,y+1,z) (x,y+1,z) (x,y,z+1) (x+1,y,z+1) (x+1,y+1,z+1) (x,y+1,z+1)
                const unsigned a0 = in[0] >> i, a1 = in[1] >> i, a2 = in[2] >> i, a3 = in[3] >> i;
                unsigned ci = (a0 & 1) | (a0 & 2) | ((a1 & 2) << 1) | ((a1 & 1) << 3) | ((a2 & 1) << 4) | ((a2 & 2) << 4) |
                              ((a3 & 2) << 5) | ((a3 & 1) << 7);
                if (i >= ncell) ci = 0;
                idx[4 * h + i] = ci;
            }
        }
    }
    // ranks in sweep order: lanes below (ballots over the bits of the per-thread count), waves below (LDS)
    bool act[CELLS_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CELLS_PER_THREAD; ++i) act[i] = false;
    const unsigned long long below = (1ull << lane) - 1ull;
    int pa = 0, wa = 0;
    if (!wave_empty) {
        int ta = 0;
#pragma unroll
        for (int i = 0; i < CELLS_PER_THREAD; ++i) {
            act[i] = idx[i] != 0u && idx[i] != 255u;
            ta += act[i];
        }
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) {   // ta <= 8
            const unsigned long long m = __ballot((ta >> bit) & 1);
            pa += __popcll(m & below) << bit;
            wa += __popcll(m) << bit;
        }
    }
    if (!FIRST) __syncthreads();   // the previous block's list has been consumed
    if (MINMAX) {
        const float wlo = wave_min_to_lane63(lo), whi = wave_max_to_lane63(hi);
        if (lane == 63) {
            list.wave_lo[wave] = wlo;
            list.wave_hi[wave] = whi;
        }
    }
    if (lane == 0) list.wave_count[wave] = wa;
    __syncthreads();
    int base = pa, n = 0;
#pragma unroll
    for (int w = 0; w < SCAN_WAVES; ++w) {
        if (w < wave) base += list.wave_count[w];
        n += list.wave_count[w];
    }
    if (n == 0) return 0;
    if (threadIdx.x == 0) list.n = n;
#pragma unroll
    for (int i = 0; i < CELLS_PER_THREAD; ++i)
        if (act[i]) {
            list.x[base] = (unsigned short)(threadIdx.x * CELLS_PER_THREAD + i);
            list.index[base] = (unsigned char)idx[i];
            ++base;
        }
    __syncthreads();
    return n;
}

// phase 2: list entry e -> its cell code and flat cell number (one entry per thread and round: the corner re-read and the
// table look-ups are a chain of two memory latencies, so a block's entries go side by side, not one after the other)
__device__ __forceinline__ unsigned mc_classify_entry(const float *__restrict__ vol, const Dims &d, double level, long long q0,
                                                      const CellList &list, int e, unsigned &cell) {
    int x, y, z;
    decode_q(d, q0 + list.x[e], x, y, z);
    cell = (unsigned)(((long long)z * d.cy + y) * d.cx + x);
    const int index = (int)list.index[e];
    const unsigned own = owned_mask(x, y, z + d.zoff);
    // the cases that need no face / interior test - every cell of a smooth surface - are settled by the cube index alone: the
    // corner values (8 scattered loads, converted to double) are fetched for the ambiguous cases only
    const unsigned fast = g_fast_code[index];
    if (fast != 0u) return fast | ((unsigned)__popc(g_fast_seen[index] & own) << 4);
    Cell c;
    float lo = 0.f, hi = 0.f;
    load_cell(vol, d, x, y, z, level, c, lo, hi);
    return classify_cell(c.v, index, own);
}

__global__ __launch_bounds__(SCAN_THREADS) void mc_count_kernel(const float *__restrict__ vol, Dims d, double level, float levelf,
                                                           int nblocks, BlockSums *__restrict__ block_counts,
                                                           float2 *__restrict__ block_minmax, int *__restrict__ nan_flag,
                                                           uint2 *__restrict__ codes) {
    __shared__ CellList list;
    __shared__ int red[3][SCAN_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // one workgroup per block: a persistent form (4 096 workgroups walking runs of blocks, one barrier more per block) was 30 % slower -
    // the blocks of a workgroup then run one after the other, each behind a full memory round trip
    {
        const int blk = (int)blockIdx.x;
        const long long q0 = d.q_begin + (long long)blk * CELLS_PER_BLOCK;
        float lo = FLT_MAX, hi = -FLT_MAX;
        const int n = mc_scan_block<true, true>(vol, d, levelf, q0, list, lo, hi, nan_flag);
        int nt_sum = 0, nv_sum = 0, na_sum = 0;
        if (n != 0) {   // uniform
            // the block's (cell, code) pairs go to the block's own 1024 slots of the scratch list, in sweep order: the emit pass
            // reads them back instead of classifying the block's cells a second time.  (One shared cursor bumped with an atomic per
            // block would pack them - and costs 0.26 ms at 512^3: 37 000 atomics on one address.)
            for (int e = threadIdx.x; e < n; e += SCAN_THREADS) {
                unsigned cell;
                const unsigned code = mc_classify_entry(vol, d, level, q0, list, e, cell);
                codes[(size_t)blk * CELLS_PER_BLOCK + e] = make_uint2(cell, code);
                nt_sum += code_nt(code);
                nv_sum += code_nv(code);
                na_sum += 1;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {   // (LDS atomics of the classifying lanes instead: the same on a smooth field, 1.4x slower on noise)
                nt_sum += __shfl_xor(nt_sum, o);
                nv_sum += __shfl_xor(nv_sum, o);
                na_sum += __shfl_xor(na_sum, o);
            }
            if (lane == 0) { red[0][wave] = nv_sum; red[1][wave] = nt_sum; red[2][wave] = na_sum; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            BlockSums bs = {0, 0, 0, 0};
            float blo = list.wave_lo[0], bhi = list.wave_hi[0];
#pragma unroll
            for (int w = 0; w < SCAN_WAVES; ++w) {
                if (n != 0) { bs.nv += red[0][w]; bs.nt += red[1][w]; bs.na += red[2][w]; }
                blo = fminf(blo, list.wave_lo[w]);
                bhi = fmaxf(bhi, list.wave_hi[w]);
            }
            block_counts[blk] = bs;
            block_minmax[blk] = make_float2(blo, bhi);
        }
    }
}

// ---------------------------------------------------------------- pass 2: exclusive scan of the block sums, two levels
// Level 1: one workgroup per 1024 blocks scans its entries (offsets local to the group) and leaves the group's totals and
// min / max; level 2 (mc_scan_kernel, one workgroup) scans the group totals.  Offset of block b = group[b >> 10] + local[b].
constexpr int SCAN_GROUP = 1024;

__global__ __launch_bounds__(1024) void mc_scan1_kernel(const BlockSums *__restrict__ counts, BlockSums *__restrict__ local,
                                                        int nblocks, BlockSums *__restrict__ group_counts,
                                                        const float2 *__restrict__ block_minmax, float2 *__restrict__ group_minmax) {
    __shared__ int wsum[3][16];
    __shared__ float wmm[2][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * SCAN_GROUP + threadIdx.x;
    BlockSums v = {0, 0, 0, 0};
    float lo = FLT_MAX, hi = -FLT_MAX;
    if (i < nblocks) {
        v = counts[i];
        const float2 mm = block_minmax[i];
        lo = mm.x;
        hi = mm.y;
    }
    int a = v.nv, b = v.nt, c = v.na;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int na = __shfl_up(a, o), nb = __shfl_up(b, o), nc = __shfl_up(c, o);
        if (lane >= o) { a += na; b += nb; c += nc; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (lane == 63) { wsum[0][wave] = a; wsum[1][wave] = b; wsum[2][wave] = c; }
    if (lane == 0) { wmm[0][wave] = lo; wmm[1][wave] = hi; }
    __syncthreads();
    int pa = 0, pb = 0, pc = 0;
    for (int w = 0; w < wave; ++w) { pa += wsum[0][w]; pb += wsum[1][w]; pc += wsum[2][w]; }
    if (i < nblocks) {
        BlockSums o;
        o.nv = pa + a - v.nv; o.nt = pb + b - v.nt; o.na = pc + c - v.na; o.pad = 0;
        local[i] = o;
    }
    if (threadIdx.x == 1023) {
        BlockSums t;
        t.nv = pa + a; t.nt = pb + b; t.na = pc + c; t.pad = 0;
        group_counts[blockIdx.x] = t;
    }
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) { lo = fminf(lo, wmm[0][w]); hi = fmaxf(hi, wmm[1][w]); }
        group_minmax[blockIdx.x] = make_float2(lo, hi);
    }
}

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

// ---------------------------------------------------------------- pass 3: emit the active cells in sweep order
// Only blocks with active cells do any work.  A thread's eight cells are consecutive in sweep order, so the rank of an
// active cell is (active cells of the lanes below, by wavefront ballots over the per-thread counts) + (its rank inside the
// thread); the running vertex / triangle numbers likewise, as sums of popcounts of per-bit ballots (no shuffles).
__global__ __launch_bounds__(SCAN_THREADS) void mc_emit_kernel(const float *__restrict__ vol, Dims d, double level, float levelf,
                                                          int nblocks, const BlockSums *__restrict__ block_counts,
                                                          const BlockSums *__restrict__ local_offsets,
                                                          const BlockSums *__restrict__ group_offsets,
                                                          ActiveCell *__restrict__ alist) {
    __shared__ CellList list;
    __shared__ int wsum[2][SCAN_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    // a workgroup walks every gridDim.x-th block and skips the empty ones (launching one workgroup per block costs more than the
    // whole pass when a few per cent of the blocks have a surface; contiguous runs of blocks left the workgroups whose run lies
    // on the surface with all the work, one block after the other)
    for (int blk = (int)blockIdx.x; blk < nblocks; blk += (int)gridDim.x) {
        if (block_counts[blk].na == 0) continue;   // uniform over the workgroup
        const long long q0 = d.q_begin + (long long)blk * CELLS_PER_BLOCK;
        float lo = 0.f, hi = 0.f;
        const int n = mc_scan_block<false, false>(vol, d, levelf, q0, list, lo, hi);
        const BlockSums lo_ = local_offsets[blk], go = group_offsets[blk / SCAN_GROUP];
        int carry_v = d.base_verts + go.nv + lo_.nv, carry_t = d.base_faces + go.nt + lo_.nt;
        const int a0 = go.na + lo_.na;   // list entries are the block's active cells, in order
        for (int e0 = 0; e0 < n; e0 += SCAN_THREADS) {   // rounds of 128 entries, one per thread
            const int e = e0 + (int)threadIdx.x;
            unsigned code = 0u, cell = 0u;
            if (e < n) code = mc_classify_entry(vol, d, level, q0, list, e, cell);
            const int tv = code_nv(code), tt = code_nt(code);
            int pv = 0, pt = 0, wv = 0, wt = 0;   // prefixes inside the wave, the wave's totals
#pragma unroll
            for (int bit = 0; bit < 4; ++bit) {   // a cell creates at most 13 vertices, 12 triangles
                const unsigned long long mv = __ballot((tv >> bit) & 1), mt = __ballot((tt >> bit) & 1);
                pv += __popcll(mv & below) << bit;
                wv += __popcll(mv) << bit;
                pt += __popcll(mt & below) << bit;
                wt += __popcll(mt) << bit;
            }
            __syncthreads();   // wsum of the previous round has been read
            if (lane == 0) { wsum[0][wave] = wv; wsum[1][wave] = wt; }
            __syncthreads();
            int run_v = carry_v + pv, run_t = carry_t + pt;
#pragma unroll
            for (int w = 0; w < SCAN_WAVES; ++w) {
                if (w < wave) { run_v += wsum[0][w]; run_t += wsum[1][w]; }
                carry_v += wsum[0][w];
                carry_t += wsum[1][w];
            }
            if (code != 0u) {
                ActiveCell ac;
                ac.cell = cell;
                ac.code = code;
                ac.vid0 = run_v;
                ac.tri0 = run_t;
                alist[(size_t)(a0 + e)] = ac;
            }
        }
        // (the next block's mc_scan_block starts with a barrier: wsum and the list are free by then)
    }
}

// Pass 3 as it normally runs: the active cells of a block were classified - and their codes stored - by the count pass; one WAVE per
// block with active cells puts them in sweep order with their running vertex / triangle numbers.  No barrier, no LDS, no look at
// the volume.  (mc_emit_kernel above, which classifies again, remains for fields where more than half of the cells are active: the
// scratch list then overlaps the sorted one.)
__global__ __launch_bounds__(THREADS) void mc_emit_list_kernel(Dims d, int nblocks, const BlockSums *__restrict__ block_counts,
                                                               const BlockSums *__restrict__ local_offsets,
                                                               const BlockSums *__restrict__ group_offsets,
                                                               const uint2 *__restrict__ codes, ActiveCell *__restrict__ alist) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int blk = (int)blockIdx.x * 4 + wave; blk < nblocks; blk += (int)gridDim.x * 4) {
        const BlockSums bc = block_counts[blk];
        if (bc.na == 0) continue;   // uniform over the wave
        const BlockSums lo_ = local_offsets[blk], go = group_offsets[blk / SCAN_GROUP];
        int carry_v = d.base_verts + go.nv + lo_.nv, carry_t = d.base_faces + go.nt + lo_.nt;
        const int a0 = go.na + lo_.na;
        for (int e0 = 0; e0 < bc.na; e0 += 64) {
            const int e = e0 + lane;
            uint2 cc = make_uint2(0u, 0u);
            if (e < bc.na) cc = codes[(size_t)blk * CELLS_PER_BLOCK + e];
            const int tv = code_nv(cc.y), tt = code_nt(cc.y);
            int pv = 0, pt = 0, wv = 0, wt = 0;   // prefixes inside the wave, the wave's totals
#pragma unroll
            for (int bit = 0; bit < 4; ++bit) {   // a cell creates at most 13 vertices, 12 triangles
                const unsigned long long mv = __ballot((tv >> bit) & 1), mt = __ballot((tt >> bit) & 1);
                pv += __popcll(mv & below) << bit;
                wv += __popcll(mv) << bit;
                pt += __popcll(mt & below) << bit;
                wt += __popcll(mt) << bit;
            }
            if (e < bc.na) {
                ActiveCell ac;
                ac.cell = cc.x;
                ac.code = cc.y;
                ac.vid0 = carry_v + pv;
                ac.tri0 = carry_t + pt;
                alist[(size_t)(a0 + e)] = ac;
            }
            carry_v += wv;
            carry_t += wt;
        }
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

// the content of MCL_EDGE_DX / DY / DZ as arithmetic: axis and lower end of each cell edge, five bits per edge
constexpr unsigned long long MC_EDGE_PACK = 0x538c28e2b00a0a0ull;   // edge e: bits 5e.. = axis (2 bits) | min dx | min dy | min dz
__device__ __forceinline__ size_t edge_slot_index(int e, int x, int y, int z, size_t nvox, int ny, int nx, int ring) {
    const unsigned v = e == 12 ? 3u : (unsigned)((MC_EDGE_PACK >> (5 * e)) & 31ull);   // centre vertex: table 3, the cell's own voxel
    const unsigned axis = v & 3u;
    const int vx = x + (int)((v >> 2) & 1u), vy = y + (int)((v >> 3) & 1u), vz = z + (int)((v >> 4) & 1u);
    return (size_t)axis * nvox + ((size_t)(vz % ring) * ny + vy) * nx + vx;
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
    const size_t nvox = (size_t)d.nx * d.ny * d.ring;
    int x, y, z;
    cell_xyz(d, (long long)ac.cell, x, y, z);
    Cell cell;
    float lo = 0, hi = 0;
    load_cell(vol, d, x, y, z, level, cell, lo, hi);
    const Tiling t = decode_cell(ac.code);
    const unsigned own = owned_mask(x, y, z + d.zoff);
    // The cell's triangle row is read in one go (all bytes at once, predicated) and reduced by arithmetic to the edges whose vertex
    // this cell creates, in order of first appearance (4 bits each); only those take the loop with the double-precision
    // interpolation.  Walking the row one byte at a time put a memory latency in front of every corner; the edge tables are
    // arithmetic (MC_EDGE_PACK, the ring structure of the corner numbers) and the corner values are picked by selects, not by
    // indexing a local array (which lived in scratch memory).
    const int n3 = 3 * t.nt;
    int eb[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) eb[i] = i < n3 ? (int)t.row[i] : 0;
    unsigned seen = 0, cnt = 0;
    unsigned long long lst = 0ull;
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        const unsigned bit = 1u << eb[i];
        const bool fresh = i < n3 && !(seen & bit);
        seen |= fresh ? bit : 0u;
        const bool mine = fresh && (own & bit);
        lst |= mine ? ((unsigned long long)eb[i] << (4 * cnt)) : 0ull;
        cnt += mine ? 1u : 0u;
    }
    const double *v = cell.v;
    auto pick = [&](int k) {   // v[k] without a dynamically indexed array
        const double a0 = (k & 1) ? v[1] : v[0], a1 = (k & 1) ? v[3] : v[2], a2 = (k & 1) ? v[5] : v[4], a3 = (k & 1) ? v[7] : v[6];
        const double b0 = (k & 2) ? a1 : a0, b1 = (k & 2) ? a3 : a2;
        return (k & 4) ? b1 : b0;
    };
    int vid = ac.vid0;
    for (unsigned j = 0; j < cnt; ++j) {
        const int e = (int)((lst >> (4 * j)) & 15ull);
        double px, py, pz;
        if (e == 12) {
            const int cx[8] = {0, 1, 1, 0, 0, 1, 1, 0}, cy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, cz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
            double fx = 0, fy = 0, fz = 0, ff = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double w = 1.0 / (MC_EPS + fabs(v[k]));
                fx += cx[k] * w; fy += cy[k] * w; fz += cz[k] * w; ff += w;
            }
            px = (double)x + fx / ff; py = (double)y + fy / ff; pz = (double)(z + d.zoff) + fz / ff;
        } else {
            // corners of edge e: 0..3 the ring of the lower face, 4..7 of the upper face, 8..11 the verticals; corner k sits at
            // (dx, dy, dz) = (((k + 1) >> 1) & 1, (k >> 1) & 1, k >> 2): v0..v7 = (0,0,0)(1,0,0)(1,1,0)(0,1,0)(0,0,1)(1,0,1)(1,1,1)(0,1,1)
            const int k1 = e < 8 ? (e & 4) + (e & 3) : e - 8, k2 = e < 8 ? (e & 4) + ((e + 1) & 3) : e - 4;
            const int dx1 = ((k1 + 1) >> 1) & 1, dy1 = (k1 >> 1) & 1, dz1 = k1 >> 2;
            const int dx2 = ((k2 + 1) >> 1) & 1, dy2 = (k2 >> 1) & 1, dz2 = k2 >> 2;
            const double w1 = 1.0 / (MC_EPS + fabs(pick(k1))), w2 = 1.0 / (MC_EPS + fabs(pick(k2)));
            double fx = 0, fy = 0, fz = 0, ff = 0;
            fx += (double)dx1 * w1; fy += (double)dy1 * w1; fz += (double)dz1 * w1; ff += w1;
            fx += (double)dx2 * w2; fy += (double)dy2 * w2; fz += (double)dz2 * w2; ff += w2;
            px = (double)x + fx / ff; py = (double)y + fy / ff; pz = (double)(z + d.zoff) + fz / ff;
        }
        evid[edge_slot_index(e, x, y, z, nvox, d.ny, d.nx, d.ring)] = vid;
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
// The row, the lattice slots and the vertex ids are fetched as in mc_face_ids_kernel below (all at once); the contributions to
// `values` and to the normals are then added once per DISTINCT edge of the cell, times the number of corners of its triangles that
// sit on that edge - not once per corner: the contribution of a corner depends on its edge only, and a cell's triangles share their
// edges (two to three corners per edge), so this is a third of the float atomics (seven per corner before: one max, six adds - the
// pass is bound by them).  Float sums in another order than the compiled core's: within the 1e-4 the normals are held to.
__global__ __launch_bounds__(THREADS) void mc_face_kernel(const float *__restrict__ vol, Dims d, double level,
                                                          const ActiveCell *__restrict__ alist, int nactive,
                                                          const int *__restrict__ evid, int *__restrict__ faces,
                                                          float *__restrict__ normals, float *__restrict__ values,
                                                          int cap_verts, int cap_faces) {
    __shared__ int lvid[13][THREADS];             // per thread: the vertex ids of the cell's distinct edges, in order of appearance
    __shared__ unsigned char ledge[13][THREADS];  // and the edges
    const int a = blockIdx.x * THREADS + threadIdx.x, tid = threadIdx.x;
    if (a >= nactive) return;
    const ActiveCell ac = alist[a];
    const size_t nvox = (size_t)d.nx * d.ny * d.ring;
    int x, y, z;
    cell_xyz(d, (long long)ac.cell, x, y, z);
    const Tiling t = decode_cell(ac.code);
    const int n3 = 3 * t.nt;
    Cell cell;
    float lo_ = 0, hi_ = 0;
    load_cell(vol, d, x, y, z, level, cell, lo_, hi_);
    int e[36], vid[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) e[i] = i < n3 ? (int)t.row[i] : 0;
#pragma unroll
    for (int i = 0; i < 36; ++i) vid[i] = i < n3 ? evid[edge_slot_index(e[i], x, y, z, nvox, d.ny, d.nx, d.ring)] : 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int tri = ac.tri0 + i;
        if (i < t.nt && tri < cap_faces) {
            // rows reversed (gradient_direction='descent')
            faces[3 * (size_t)tri + 0] = vid[3 * i + 2];
            faces[3 * (size_t)tri + 1] = vid[3 * i + 1];
            faces[3 * (size_t)tri + 2] = vid[3 * i + 0];
        }
    }
    // corners per edge (a nibble each) and the distinct edges with their vertex ids
    unsigned long long per_edge = 0ull;
    unsigned seen = 0, cnt = 0;
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        const unsigned bit = 1u << e[i];
        per_edge += i < n3 ? (1ull << (4 * e[i])) : 0ull;
        if (i < n3 && !(seen & bit)) {
            seen |= bit;
            lvid[cnt][tid] = vid[i];
            ledge[cnt][tid] = (unsigned char)e[i];
            ++cnt;
        }
    }
    const double *v = cell.v;
    // per-cell quantities for values / normals (see oracle/mc_oracle.c for the black-box verified quirks)
    double vlo = 0.0, vhi = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (v[k] > vhi) vhi = v[k]; if (v[k] < vlo) vlo = v[k]; }
    const float vrange = (float)(vhi - vlo);
    const double g[8][3] = {
        {v[0] - v[1], v[0] - v[3], v[0] - v[4]}, {v[0] - v[1], v[1] - v[2], v[1] - v[5]},
        {v[3] - v[2], v[1] - v[2], v[2] - v[6]}, {v[3] - v[2], v[0] - v[3], v[3] - v[7]},
        {v[4] - v[5], v[4] - v[7], v[0] - v[4]}, {v[4] - v[5], v[5] - v[6], v[1] - v[5]},
        {v[7] - v[6], v[5] - v[6], v[2] - v[6]}, {v[7] - v[6], v[4] - v[7], v[3] - v[7]}};
    auto pickv = [&](int k) {   // v[k], g[k][ax] without dynamically indexed arrays (they would live in scratch memory)
        const double a0 = (k & 1) ? v[1] : v[0], a1 = (k & 1) ? v[3] : v[2], a2 = (k & 1) ? v[5] : v[4], a3 = (k & 1) ? v[7] : v[6];
        const double b0 = (k & 2) ? a1 : a0, b1 = (k & 2) ? a3 : a2;
        return (k & 4) ? b1 : b0;
    };
    auto pickg = [&](int k, int ax) {
        const double a0 = (k & 1) ? g[1][ax] : g[0][ax], a1 = (k & 1) ? g[3][ax] : g[2][ax], a2 = (k & 1) ? g[5][ax] : g[4][ax],
                     a3 = (k & 1) ? g[7][ax] : g[6][ax];
        const double b0 = (k & 2) ? a1 : a0, b1 = (k & 2) ? a3 : a2;
        return (k & 4) ? b1 : b0;
    };
    double c12g[3] = {0, 0, 0};
    if (normals) {
        double gy = 0, gz = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double w = 1.0 / (MC_EPS + fabs(v[k]));
            gy += w * g[k][1];
            gz += w * g[k][2];
        }
        c12g[0] = gz; c12g[1] = gy; c12g[2] = 0.0;  // quirk of the compiled core: (Gz, Gy, 0)
    }
    for (unsigned j = 0; j < cnt; ++j) {
        const int ed = (int)ledge[j][tid], vj = lvid[j][tid];
        if (vj >= cap_verts) continue;
        const float mult = (float)((per_edge >> (4 * ed)) & 15ull);   // corners of the cell's triangles on this edge
        if (values) atomicMax(reinterpret_cast<int *>(values) + vj, __float_as_int(vrange));
        if (normals) {
            float *n = normals + 3 * (size_t)vj;
            if (ed == 12) {
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) atomicAdd(n + ax, mult * (float)c12g[ax]);
            } else {
                // corners of edge ed (see mc_vertex_kernel); the core indexes its corner-number gradient table with the xyz-bit
                // index of the corner (quirk)
                const int k1 = ed < 8 ? (ed & 4) + (ed & 3) : ed - 8, k2 = ed < 8 ? (ed & 4) + ((ed + 1) & 3) : ed - 4;
                const int i1 = (k1 >> 2) * 4 + ((k1 >> 1) & 1) * 2 + (((k1 + 1) >> 1) & 1);
                const int i2 = (k2 >> 2) * 4 + ((k2 >> 1) & 1) * 2 + (((k2 + 1) >> 1) & 1);
                const double w1 = 1.0 / (MC_EPS + fabs(pickv(k1))), w2 = 1.0 / (MC_EPS + fabs(pickv(k2)));
#pragma unroll
                for (int ax = 0; ax < 3; ++ax)
                    atomicAdd(n + ax, mult * ((float)(pickg(i1, ax) * w1) + (float)(pickg(i2, ax) * w2)));
            }
        }
    }
}

// ---------------------------------------------------------------- pass 4b without normals / values (what gen_mesh asks for)
// mc_face_kernel walks a cell's triangle list one corner at a time: table byte -> six bytes of the edge tables -> vertex id, three
// dependent memory latencies per corner and up to 36 corners - a latency chain, which is what the pass costs (one thread per active
// cell, everything resident at once).  Here the chain is four loads deep whatever the cell: all bytes of the triangle row at once
// (predicated), the lattice slot of every corner by arithmetic (MC_EDGE_PACK: axis and lower end of each cell edge, five bits per
// edge - the content of MCL_EDGE_DX / DY / DZ), all vertex ids at once, then the stores.
__global__ __launch_bounds__(THREADS) void mc_face_ids_kernel(Dims d, const ActiveCell *__restrict__ alist, int nactive,
                                                              const int *__restrict__ evid, int *__restrict__ faces, int cap_faces) {
    const int a = blockIdx.x * THREADS + threadIdx.x;
    if (a >= nactive) return;
    const ActiveCell ac = alist[a];
    const size_t nvox = (size_t)d.nx * d.ny * d.ring;
    int x, y, z;
    cell_xyz(d, (long long)ac.cell, x, y, z);
    const Tiling t = decode_cell(ac.code);
    const int n3 = 3 * t.nt;
    int e[36], vid[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) e[i] = i < n3 ? (int)t.row[i] : 0;
#pragma unroll
    for (int i = 0; i < 36; ++i) vid[i] = i < n3 ? evid[edge_slot_index(e[i], x, y, z, nvox, d.ny, d.nx, d.ring)] : 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int tri = ac.tri0 + i;
        if (i < t.nt && tri < cap_faces) {
            // rows reversed (gradient_direction='descent')
            faces[3 * (size_t)tri + 0] = vid[3 * i + 2];
            faces[3 * (size_t)tri + 1] = vid[3 * i + 1];
            faces[3 * (size_t)tri + 2] = vid[3 * i + 0];
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

// The edge -> vertex-id tables are a RING of planes: Lewiner's sweep has axis 0 outermost, a cell layer reads and writes the ids of
// its two planes only, and every extraction proceeds in contiguous layer ranges - so a range of L layers needs L + 1 planes, and the
// plane it shares with the next range survives as long as L + 1 <= ring.  (The streamed extraction advances by at most
// 64 layers - one launch of the sweep at 512^3); larger ranges, the one-piece extraction included, are walked in chunks of ring - 1
// layers (mc_chunked).  Until round 4 the tables were dense: 4 x int32[n0 n1 n2] = 2.1 GB per field at 512^3 (now 277 MB), and the
// active-cell list was sized for every cell of the volume (2.1 GB; now for the cells of one chunk, 272 MB).
// Round 6: the ring's size is the library option mc_ring (default 258 planes: a one-piece extraction of a 512-layer volume - the
// octree mode's, a first reconstruction's - is two chunks instead of eight: 10 launches and 2 host synchronisations instead of 40 and
// 8; 2.2 GB of workspace per field at 512^3 instead of 0.55 - of 288).  Read at every call: every call on one workspace must see the
// same value (set it before the first extraction).
static int mc_ring(int n0) {
    int r = option(OPT_MC_RING);
    if (r < 3) r = 3;
    return n0 < r ? n0 : r;
}
static long long mc_chunk_cells(int n0, int n1, int n2) {   // the most cells one mc_range call processes
    const int layers = (n0 - 1) < (mc_ring(n0) - 1) ? (n0 - 1) : (mc_ring(n0) - 1);
    return (long long)layers * (n1 - 1) * (n2 - 1);
}
static size_t mc_chunk_blocks(int n0, int n1, int n2) {     // ... and the blocks of padded rows they form
    const int layers = (n0 - 1) < (mc_ring(n0) - 1) ? (n0 - 1) : (mc_ring(n0) - 1);
    return (size_t)mc_nblocks((long long)layers * (n1 - 1) * ((n2 - 1 + 3) / 4 * 4));
}

static size_t mc_ws_layout(int n0, int n1, int n2, size_t off[5]) {
    const size_t nb = mc_chunk_blocks(n0, n1, n2);   // padded rows (mc_scan_block)
    const size_t nvox = (size_t)mc_ring(n0) * n1 * n2;
    size_t o = 0;
    const size_t ng = (nb + SCAN_GROUP - 1) / SCAN_GROUP;
    off[0] = o; o += align_up(nb * sizeof(BlockSums), 256);         // block sums
    off[1] = o; o += align_up(nb * sizeof(BlockSums), 256);         // block offsets (local to their scan group)
    off[2] = o; o += 256 + align_up(nb * sizeof(float2), 256);      // min/max, totals; per-block min/max
    off[3] = o; o += 3 * align_up(ng * sizeof(BlockSums), 256);     // scan groups: totals, offsets, min/max
    off[4] = o; o += align_up(4 * nvox * sizeof(int), 256);         // edge -> vertex id tables (ring of planes)
    return o;   // the active-cell list follows (worst case: every cell of a chunk)
}

extern "C" size_t surs_mc_workspace_bytes(int n0, int n1, int n2) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return 0;
    size_t off[5];
    const long long ncells = mc_chunk_cells(n0, n1, n2);
    // the active-cell list (worst case: every cell of a chunk); the count pass's codes use its upper half, 1024 slots per block of
    // PADDED cells
    const size_t slots = mc_chunk_blocks(n0, n1, n2) * CELLS_PER_BLOCK;
    return mc_ws_layout(n0, n1, n2, off) + align_up((size_t)ncells * 8 + (slots > (size_t)ncells ? slots : (size_t)ncells) * 8, 256);
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
    d.ring = mc_ring(n0);
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
    // the classification passes walk padded rows (see mc_scan_block); ranges are whole layers, so they map to whole rows
    d.prow = (d.cx + 3) / 4 * 4;
    const long long per_layer = (long long)d.cy * d.cx;
    SURS_REQUIRE(cell_begin % per_layer == 0 && cell_end % per_layer == 0, "cell range must consist of whole layers");
    const long long list_cells = mc_chunk_cells(n0, n1, n2);   // capacity of the active-cell list = the cells of the largest range
    SURS_REQUIRE(cell_end - cell_begin <= list_cells, "range of more than ring - 1 cell layers (mc_chunked splits them)");
    d.q_begin = cell_begin / d.cx * d.prow;
    d.q_end = cell_end / d.cx * d.prow;
    {
        auto magic = [](unsigned dv, unsigned &m, unsigned &sh) {   // dv >= 2
            unsigned L = 0;
            while ((1ull << L) < dv) ++L;
            sh = L - 1;
            m = (unsigned)(((1ull << (31 + L)) + dv - 1) / dv);
        };
        d.fastdiv = d.prow >= 2 && d.cy >= 2 && (long long)d.cz * d.cy * d.prow < (1ll << 31);
        d.m_prow = d.s_prow = d.m_cy = d.s_cy = 0;
        if (d.fastdiv) {
            magic((unsigned)d.prow, d.m_prow, d.s_prow);
            magic((unsigned)d.cy, d.m_cy, d.s_cy);
        }
    }
    SURS_REQUIRE((long long)d.cz * d.cy * d.prow < (1ll << 32), "volume too large");
    const int nb = mc_nblocks(d.q_end - d.q_begin);
    {
        // the table of cell codes that need no test, filled once per device.  Under a mutex: a second host thread (the pipelined flows
        // drive two streams; surs_query_grid_opt is documented as thread-safe) that arrives while the first is still filling it must
        // wait for the COMPLETE table - first() alone would let it through to classify against a half-written one
        static std::mutex fast_mu;
        static DeviceOnce fast_table;
        std::lock_guard<std::mutex> lock(fast_mu);
        if (fast_table.first()) {
            hipLaunchKernelGGL(mc_fast_init_kernel, dim3(1), dim3(256), 0, st);
            SURS_LAUNCH_CHECK();
            SURS_HIP_CHECK(hipStreamSynchronize(st));
        }
    }
    size_t off[5];
    const size_t fixed = mc_ws_layout(n0, n1, n2, off);
    char *ws = (char *)workspace;
    BlockSums *bcounts = (BlockSums *)(ws + off[0]);
    BlockSums *boffs = (BlockSums *)(ws + off[1]);
    float *minmax = (float *)(ws + off[2]);
    int *totals = (int *)(ws + off[2] + 16);
    float2 *bminmax = (float2 *)(ws + off[2] + 256);
    const int ng = (nb + SCAN_GROUP - 1) / SCAN_GROUP;
    const size_t gstride = align_up((size_t)ng * sizeof(BlockSums), 256);
    BlockSums *gcounts = (BlockSums *)(ws + off[3]);
    BlockSums *goffs = (BlockSums *)(ws + off[3] + gstride);
    float2 *gminmax = (float2 *)(ws + off[3] + 2 * gstride);
    int *evid = (int *)(ws + off[4]);
    ActiveCell *alist = (ActiveCell *)(ws + fixed);

    // largest float <= level: for a float f, f > levelf  <=>  (double)f - level > 0 (how the core decides "inside")
    float levelf = (float)level;
    if ((double)levelf > level) levelf = nextafterf(levelf, -INFINITY);
    int *nan_flag = totals + 3;
    SURS_HIP_CHECK(hipMemsetAsync(nan_flag, 0, sizeof(int), st));
    // the count pass's (cell, code) pairs, 1024 slots per block: the upper half of the worst-case active-cell list (8 of 16 bytes
    // per cell) and the row padding behind it (surs_mc_workspace_bytes)
    uint2 *codes = (uint2 *)(ws + fixed + (size_t)list_cells * 8);
    hipLaunchKernelGGL(mc_count_kernel, dim3(nb), dim3(SCAN_THREADS), 0, st, vol, d, level, levelf, nb, bcounts, bminmax, nan_flag, codes);
    SURS_LAUNCH_CHECK();
    hipLaunchKernelGGL(mc_scan1_kernel, dim3(ng), dim3(1024), 0, st, bcounts, boffs, nb, gcounts, bminmax, gminmax);
    SURS_LAUNCH_CHECK();
    hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, st, gcounts, goffs, ng, totals, gminmax, minmax);
    SURS_LAUNCH_CHECK();
    struct { float mm[2]; unsigned pad[2]; int tot[3]; int nan; } host;
    SURS_HIP_CHECK(hipMemcpyAsync(&host, minmax, sizeof(host), hipMemcpyDeviceToHost, st));
    SURS_HIP_CHECK(hipStreamSynchronize(st));
    if (host.nan) {   // reported through the range: vmin = NaN (sticky: fminf(NaN, x) below would drop it)
        run->vmin = NAN;
        run->vmax = NAN;
        return fail(SURS_E_NONFINITE, "the volume contains NaN values");
    }
    if (run->vmin != run->vmin) return fail(SURS_E_NONFINITE, "the volume contains NaN values");
    run->vmin = fminf(run->vmin, host.mm[0]);
    run->vmax = fmaxf(run->vmax, host.mm[1]);
    const int nactive = host.tot[2];
    run->n_verts = d.base_verts + host.tot[0];
    run->n_faces = d.base_faces + host.tot[1];
    if (count_only || nactive == 0) return 0;
    if (run->n_verts > cap_verts || run->n_faces > cap_faces)
        return fail(SURS_E_CAPACITY, "output capacity too small: need %d vertices, %d faces", run->n_verts, run->n_faces);
    const int emit_reclassify = option(OPT_MC_EMIT_RECLASSIFY);   // tests
    if (2ll * nactive > list_cells || emit_reclassify)   // the sorted list would run into the count pass's codes: classify again
        hipLaunchKernelGGL(mc_emit_kernel, dim3(nb < 16384 ? nb : 16384), dim3(SCAN_THREADS), 0, st, vol, d, level, levelf, nb, bcounts, boffs,
                           goffs, alist);
    else
        hipLaunchKernelGGL(mc_emit_list_kernel, dim3(ceil_div(nb, 4) < 4096 ? ceil_div(nb, 4) : 4096), dim3(THREADS), 0, st, d, nb, bcounts,
                           boffs, goffs, codes, alist);
    SURS_LAUNCH_CHECK();
    const int ab = ceil_div(nactive, THREADS);
    hipLaunchKernelGGL(mc_vertex_kernel, dim3(ab), dim3(THREADS), 0, st, vol, d, level, alist, nactive, evid, verts, normals,
                       values, cap_verts);
    SURS_LAUNCH_CHECK();
    if (!normals && !values)
        hipLaunchKernelGGL(mc_face_ids_kernel, dim3(ab), dim3(THREADS), 0, st, d, alist, nactive, evid, faces, cap_faces);
    else
        hipLaunchKernelGGL(mc_face_kernel, dim3(ab), dim3(THREADS), 0, st, vol, d, level, alist, nactive, evid, faces, normals, values,
                           cap_verts, cap_faces);
    SURS_LAUNCH_CHECK();
    return 0;
}

// [cell_begin, cell_end) in chunks of at most ring - 1 layers (the vertex-id ring holds ring planes).  A chunk that does not fit the
// output buffers turns the rest of the walk into a counting walk: run ends at the sizes the whole range needs, SURS_E_CAPACITY.
static int mc_chunked(const float *vol, int n0, int n1, int n2, long long cell_begin, long long cell_end, double level,
                      void *workspace, size_t workspace_bytes, float *verts, float *normals, float *values, int cap_verts,
                      int32_t *faces, int cap_faces, bool count_only, surs_mc_counts *run, hipStream_t st, int zoff = 0) {
    const long long per_layer = (long long)(n1 - 1) * (n2 - 1);
    const long long step = (long long)((n0 - 1) < (mc_ring(n0) - 1) ? (n0 - 1) : (mc_ring(n0) - 1)) * per_layer;
    bool overflow = false;
    for (long long a = cell_begin; a < cell_end; a += step) {
        const long long b = a + step < cell_end ? a + step : cell_end;
        const int rc = mc_range(vol, n0, n1, n2, a, b, level, workspace, workspace_bytes, verts, normals, values, cap_verts, faces, cap_faces,
                                count_only || overflow, run, st, zoff);
        if (rc == SURS_E_CAPACITY) overflow = true;
        else if (rc) return rc;
    }
    if (overflow) return fail(SURS_E_CAPACITY, "output capacity too small: need %d vertices, %d faces", run->n_verts, run->n_faces);
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
    int rc = mc_chunked(vol, n0, n1, n2, 0, ncells, level, workspace, workspace_bytes, verts, normals, values, cap_verts, faces,
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
    return mc_chunked(vol, n0, n1, n2, layer_begin * per_layer, layer_end * per_layer, level, workspace, workspace_bytes, verts,
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
        hipLaunchKernelGGL(mc_slab_refs_kernel, dim3(ceil_div(2 * plane, 256)), dim3(256), 0, st, evid, (size_t)mc_ring(n0) * n1 * n2, plane);
        SURS_LAUNCH_CHECK();
    }
    const long long per_layer = (long long)(n1 - 1) * (n2 - 1);
    return mc_chunked(vol, n0, n1, n2, layer_begin * per_layer, layer_end * per_layer, level, workspace, workspace_bytes, verts,
                      nullptr, nullptr, cap_verts, faces, cap_faces, false, run, st, z_offset);
}

extern "C" int surs_mc_slab_top_ids(const void *workspace, size_t workspace_bytes, int n0, int n1, int n2, int32_t *ids, void *stream) {
    SURS_REQUIRE(workspace && ids && n0 >= 2 && n1 >= 2 && n2 >= 2, "bad argument");
    SURS_REQUIRE(workspace_bytes >= surs_mc_workspace_bytes(n0, n1, n2), "workspace too small");
    size_t off[5];
    mc_ws_layout(n0, n1, n2, off);
    const int *evid = (const int *)((const char *)workspace + off[4]);
    const size_t nvox = (size_t)mc_ring(n0) * n1 * n2, plane = (size_t)n1 * n2;   // (the top plane's slot in the ring of planes)
    for (int axis = 0; axis < 2; ++axis)
        SURS_HIP_CHECK(hipMemcpyAsync(ids + axis * plane, evid + axis * nvox + (size_t)((n0 - 1) % mc_ring(n0)) * plane, plane * sizeof(int),
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
