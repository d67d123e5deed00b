"""Acceptance of the reduced-precision sweeps (BASELINE configs[2] bf16, configs[4] fp16) at FULL size: 512^3 grid,
full-size feature maps (256^2 x 256 and 1024^2 x 64), against the fp32-grade sweep (column kernel v11, which the other
GPU tests hold to the reference's goldens at 1e-4) - in logit space, as a count of voxels on the other side of the 0.5
level, and on the extracted meshes (vertex / face counts, symmetric nearest-vertex distance in voxel units).  Two fields:
the bench's noise-like field and a smooth closed body-sized blob (tools/precision_report.py, SURVEY.md section 7 "parity
under reduced precision ... and a mesh-level metric").  Bounds = at most twice the values measured on MI355X
(profiles/r02_precision_report_512.json; the 99.9th-percentile distance is bounded by the matching radius)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))

#            max|dlogit| mean|dlogit| flipped fraction  |dV|/V   mean dist  p99.9 dist  unmatched fraction
BOUNDS = {
    ("body", "bf16"): (0.125, 0.021, 2.2e-4, 1.3e-4, 0.14, 0.95, 1e-5),
    ("body", "fp16"): (0.015, 0.0027, 2.8e-5, 4e-5, 0.023, 0.50, 1e-5),
    ("noise", "bf16"): (0.0113, 0.002, 1.9e-3, 5e-3, 0.044, 0.95, 1.2e-3),
    ("noise", "fp16"): (0.0015, 0.00023, 1.1e-4, 1.3e-4, 0.004, 0.22, 1e-5),
}


@pytest.fixture(scope="module")
def reports():
    import precision_report as pr
    from surs_amd import native
    dev = native.require_gpu()
    out = {}
    sd, Fl, Fh = pr.body_inputs(dev)
    out["body"] = pr.report(sd, Fl, Fh, 512, dev)
    del Fl, Fh
    sd, Fl, Fh, keep = pr.noise_inputs(dev)
    out["noise"] = pr.report(sd, Fl, Fh, 512, dev)
    return out


@pytest.mark.parametrize("field,prec", sorted(BOUNDS))
def test_reduced_precision_acceptance_512(reports, field, prec):
    b = BOUNDS[(field, prec)]
    rep = reports[field][prec]
    for tag in ("hr", "lr"):
        r = rep[tag]
        m = r["mesh"]
        print(field, prec, tag, {k: r[k] for k in ("max_abs_dlogit", "mean_abs_dlogit", "flipped_voxels")}, m)
        assert r["max_abs_dlogit"] < b[0] and r["mean_abs_dlogit"] < b[1], (field, prec, tag, r)
        assert r["flipped_fraction"] < b[2], (field, prec, tag, r["flipped_voxels"])
        assert abs(m["verts"] - m["verts_ref"]) <= b[3] * m["verts_ref"] and abs(m["faces"] - m["faces_ref"]) <= b[3] * m["faces_ref"]
        for side in ("to_ref", "from_ref"):
            d = m[side]
            assert d["mean"] < b[4] and d["p999"] < b[5], (field, prec, tag, side, d)
            assert d["unmatched"] <= max(2, b[6] * d["n"]), (field, prec, tag, side, d)
    # the fp32-grade sweep of the whole 512^3 grid stays under the north star's 2 s
    assert reports[field]["sweep_s"]["fp32"] < 2.0


def test_body_field_is_one_closed_surface(reports):
    """The `body` field is what it claims: a closed genus-0 surface (F = 2V - 4) of body-like size, away from the grid border."""
    m = reports["body"]["fp16"]["hr"]["mesh"]
    assert m["faces_ref"] == 2 * m["verts_ref"] - 4
    assert 2e5 < m["verts_ref"] < 1e6
