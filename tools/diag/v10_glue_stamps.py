"""Adds accumulating phase stamps around the glue of grid_mlp_kernel_v10 (column fetch, constants, tile prologue / epilogue) to a working
copy - diagnostic only (the line shifts change the library hash: do not commit the result).  Then: tools/build_trace.sh;
SURS_V3_TRACE=1 SURS_LIB_PATH=abl/libsurs_trace.so python tools/gpu_grid_once_trace.py bf16 512"""
p='/root/repo/super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd/csrc/surs_grid_v10.inc'
s=open(p).read()
def rep(old,new,count=1):
    global s
    assert s.count(old)==count,(s.count(old),old[:70])
    s=s.replace(old,new)
rep("""#ifdef SURS_V3_TRACE
    int ncol_done = 0;
#endif
    for (;;) {
        if (tid == 0) *colslot = (int)atomicAdd(a.colctr, 1u);
        __syncthreads();  // everyone is done with the previous column's constants; the next column's index is there
        const int col = *colslot;
        if (col >= a.ncols) break;""","""#ifdef SURS_V3_TRACE
    int ncol_done = 0;
    // where a workgroup's time goes outside the two classifiers (workgroup 0, thread 0; sums over the launch):
    // [50] column fetch (atomic + barrier), [51] constants (loads + barrier), [52] tile prologue (z values), [53] lr classifier,
    // [54] sigmoid between, [55] hr classifier, [56] sigmoid + stores
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define V10_ACC(i) do { if (tid == 0 && blockIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; } } while (0)
#else
#define V10_ACC(i) do { } while (0)
#endif
    for (;;) {
        V10_ACC(6);
        if (tid == 0) *colslot = (int)atomicAdd(a.colctr, 1u);
        __syncthreads();  // everyone is done with the previous column's constants; the next column's index is there
        const int col = *colslot;
        V10_ACC(0);
        if (col >= a.ncols) break;""")
rep("""        const unsigned short *kl = a.klist ? a.klist + (size_t)col * a.rz : nullptr;
        __syncthreads();
        for (int zc = 0; zc < nzc; ++zc) {
            if (zc * 128 >= cnt) break;""","""        const unsigned short *kl = a.klist ? a.klist + (size_t)col * a.rz : nullptr;
        __syncthreads();
        V10_ACC(1);
        for (int zc = 0; zc < nzc; ++zc) {
            if (zc * 128 >= cnt) break;
            V10_ACC(6);""")
rep("""            const float zlo = fminf(ze0, ze1), zhi = fmaxf(ze0, ze1);
            grid_mlp_v10<DT, 0>(w_lr, src, a.zvec, w1t, rv_lr, a.rld_lr, a.zmid, zlo, zhi, smem, wave, lane, zf, p0, l, ksteps);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) p_lr[ct] = cmask * (1.0f / (1.0f + expf(-l[ct])));
            grid_mlp_v10<DT, 1>(w_hr, src, a.zvec, w1t + (size_t)D1 * D2, rv_hr, a.rld_hr, a.zmid, zlo, zhi, smem, wave, lane, zf, p_lr, l, ksteps);""","""            const float zlo = fminf(ze0, ze1), zhi = fmaxf(ze0, ze1);
            V10_ACC(2);
            grid_mlp_v10<DT, 0>(w_lr, src, a.zvec, w1t, rv_lr, a.rld_lr, a.zmid, zlo, zhi, smem, wave, lane, zf, p0, l, ksteps);
            V10_ACC(3);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) p_lr[ct] = cmask * (1.0f / (1.0f + expf(-l[ct])));
            V10_ACC(4);
            grid_mlp_v10<DT, 1>(w_hr, src, a.zvec, w1t + (size_t)D1 * D2, rv_hr, a.rld_hr, a.zmid, zlo, zhi, smem, wave, lane, zf, p_lr, l, ksteps);
            V10_ACC(5);""")
rep("""    if (tid == 0 && blockIdx.x == 0) {
        g_v3_trace[42] = __builtin_amdgcn_s_memtime();
        g_v3_trace[43] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}""","""    if (tid == 0 && blockIdx.x == 0) {
        g_v3_trace[42] = __builtin_amdgcn_s_memtime();
        g_v3_trace[43] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 7; ++i) g_v3_trace[50 + i] = tacc[i];
        g_v3_trace[57] = (unsigned long long)ncol_done;
    }
#endif
#undef V10_ACC
}""")
open(p,'w').write(s)
p='/root/repo/super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd/csrc/surs_query.hip'
s=open(p).read()
rep("""        fprintf(stderr, "v3 trace between MLPs: %llu\\n", t[16] - t[9]);""","""        fprintf(stderr, "v3 trace between MLPs: %llu\\n", t[16] - t[9]);
        if (t[57])
            fprintf(stderr, "workgroup 0 over %llu columns, cycles per column: fetch %llu, constants %llu; per tile: prologue %llu, lr %llu, between %llu, hr %llu, "
                    "epilogue %llu\\n", t[57], t[50] / t[57], t[51] / t[57], t[52] / (4 * t[57]), t[53] / (4 * t[57]), t[54] / (4 * t[57]), t[55] / (4 * t[57]),
                    t[56] / (4 * t[57]));""")
open(p,'w').write(s)
