"""The compiler hazards found on hardware (NOTES R5.1, R5.7), held in the SHIPPED ISA: this test disassembles the gfx950 code
objects of libsurs_hip.so (llvm-objdump / llvm-readelf of /opt/rocm/lib/llvm: CPU only) and fails if hipcc has re-formed

* ``v_pk_fma_f32 .. op_sel:[0,1,0]`` anywhere in the library - on MI355X the low half of that instruction lost its product in
  lanes 48 - 63 when two workgroups shared a CU (kernel v12, layer 4's tail; pinned by SURS_ISA_PIN in surs_grid_v12.inc);
* ``v_fma_mix{lo,hi}_f16`` in the restated column kernels v10 / v12 - a fold of the residual product and its conversion into a
  single rounding, which hipcc applied to one element in eight and which the v12 == v10 bit-for-bit contract cannot tolerate;
* a kernel that needs more unified registers than its residency plan has (two workgroups - or eight waves - per CU: 256), or a
  NEW kernel beyond 256 unified registers that has not been run under GPU sharing (tests/test_gpu_dist.py) and listed here;
* scratch (register spills) in the column kernels beyond what is recorded here.

SURS_ISA_SO=<path> points the test at another build; a library built with -DSURS_ABL_NO_ISA_PINS (tools/dev/build_variant.sh nopins surs_query.hip
-DSURS_ABL_NO_ISA_PINS) makes the first two tests fail - checked in round 6 (NOTES R6.1)."""
import os
import re

import pytest

import isa

pytestmark = pytest.mark.skipif(not isa.available(), reason="llvm-objdump / llvm-readelf of /opt/rocm/lib/llvm or the built library missing")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    so = os.environ.get("SURS_ISA_SO", isa.SO)
    out = {}
    for co in isa.code_objects(so, str(tmp_path_factory.mktemp("isa"))):
        md, dis = isa.kernel_metadata(co), isa.disassembly(co)
        names = isa.demangle(list(md))
        for k, m in md.items():
            assert k in dis and dis[k], "no disassembly for kernel %s" % k
            out[names[k]] = (m, dis[k])
    assert len(out) >= 90, "expected the library's ~ 100 kernels, found %d" % len(out)
    return out


def _col(kernels, *versions):
    pat = re.compile(r"grid_mlp_kernel_(%s)\b" % "|".join(versions))
    sel = {n: v for n, v in kernels.items() if pat.search(n)}
    assert sel, "no column kernel %s in the library" % (versions,)
    return sel


def test_no_packed_fma_with_the_operand_selection_that_lost_a_product(kernels):
    bad = []
    for name, (_, ins) in kernels.items():
        bad += ["%s: %s" % (name[:60], i) for i in ins if i.startswith("v_pk_fma_f32") and "op_sel:[0,1,0]" in i]
    assert not bad, "v_pk_fma_f32 op_sel:[0,1,0] re-formed (NOTES R5.1):\n" + "\n".join(bad[:10])
    # the check looks at real code: the column kernels do contain packed fmas (other operand selections)
    assert any(i.startswith("v_pk_fma_f32") for _, ins in _col(kernels, "v12").values() for i in ins)


def test_residual_products_are_rounded_to_fp32_before_the_16_bit_conversion(kernels):
    bad = []
    for name, (_, ins) in _col(kernels, "v10", "v12").items():
        bad += ["%s: %s" % (name[:60], i) for i in ins if i.startswith("v_fma_mix")]
    assert not bad, "v_fma_mix* folds in the restated column kernels (NOTES R5.1):\n" + "\n".join(bad[:10])


# kernels planned for two workgroups (or eight waves) per CU: at most 256 unified registers (VGPR + AGPR) per lane
CO_RESIDENT = [r"grid_mlp_kernel_v12<", r"grid_mlp_kernel_v10<", r"grid_mlp_kernel_v11\b", r"conv_x3_kernel<3, 1, [48], 32, [12]>", r"conv_x3_kernel<3, 1, 8, 64, 2>",
               r"conv1x1_x2_kernel<", r"gemm_x3g_kernel<"]
# kernels that are ALLOWED beyond 256 (one workgroup of four waves per CU by design), with the register count they shipped with when
# they last passed tests/test_gpu_dist.py (four processes on one GPU: waves preempted mid-kernel, NOTES R5.7)
BEYOND_256 = {
    r"grid_mlp_kernel_v3<": 512, r"grid_mlp_kernel_v5\b": 512,
    r"conv_x3_kernel<3, 2, 4, 32, [12]>": 304, r"conv_x3_kernel<3, 1, 4, 32, 3>": 288,
    r"rvec_small_kernel<3>": 288,   # (three bf16 parts: the wide-operand retry's; its prefetch ring of fragments)
}


def _regs(m):
    return int(m[".vgpr_count"])   # gfx90a+: the unified count (arch VGPRs rounded up + AGPRs)


def test_register_budgets(kernels):
    for pat in CO_RESIDENT:
        sel = [(n, _regs(m)) for n, (m, _) in kernels.items() if re.search(pat, n)]
        assert sel, "no kernel matches %s" % pat
        for n, r in sel:
            assert r <= 256, "%s: %d unified registers, its residency plan has 256" % (n[:70], r)
    for n, (m, _) in kernels.items():
        r = _regs(m)
        assert int(m[".agpr_count"]) <= r <= 512
        if r > 256:
            lim = [v for p, v in BEYOND_256.items() if re.search(p, n)]
            assert lim, ("%s: %d unified registers - a new kernel beyond 256: run tests/test_gpu_dist.py with it (NOTES R5.7) and "
                         "list it in BEYOND_256" % (n[:70], r))
            assert r <= lim[0], "%s: %d unified registers, listed with %d" % (n[:70], r, lim[0])


# scratch bytes per lane the column kernels ship with (profiles/pmc_summary.json: scratch_bytes_per_lane); lower is fine, more is not
SCRATCH = {r"grid_mlp_kernel_v12<": 0, r"grid_mlp_kernel_v10<": 0, r"grid_mlp_kernel_v11\b": 0, r"grid_mlp_kernel_v3<": 28,
           r"grid_mlp_kernel_v5\b": 0}


def test_scratch_of_the_column_kernels(kernels):
    for pat, lim in SCRATCH.items():
        for n, (m, ins) in kernels.items():
            if re.search(pat, n):
                sb = int(m[".private_segment_fixed_size"])
                assert sb <= lim, "%s: %d bytes of scratch per lane (recorded: %d)" % (n[:70], sb, lim)
                if lim == 0:
                    # (v12, round 6: the output pointer and `lane & 31`, derived from the lane index at the START of an item and needed at
                    #  its END, had been spilled - a scratch store per item, 0.4 GB of HBM writes per launch; they are made on the spot
                    #  now: fresh_lane() in surs_query.hip)
                    assert not any(i.startswith("scratch_") for i in ins)
