"""CPU suite: the C-ABI library loads, exports every symbol include/strelka_amd.h declares, and fails loudly (no silent
CPU fallback) when there is no gfx950 device."""
import os
import re

import numpy as np
import pytest

from strelka_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "strelka_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(built):
    L = capi.lib()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libstrelka_amd.so does not export %s" % n
    assert set(capi.EXPORTS) <= set(names)


def test_version_and_defaults(built):
    assert capi.lib().sk_version() == 100
    g = capi.germline_options()
    assert (g.bsnp_diploid_theta, g.bsnp_ssd_no_mismatch, g.bsnp_ssd_one_mismatch, g.is_min_vexp, g.min_vexp) == \
        (0.001, 0.35, 0.6, 1, 0.25)
    s = capi.somatic_snv_options()
    assert (s.somatic_snv_rate, s.shared_site_error_rate, s.ssnv_contam_tolerance) == (1e-4, 5e-10, 0.15)


def test_struct_sizes(built):
    import ctypes as C
    assert C.sizeof(capi.ScoreOp) == 8 and capi.SCORE_OP_DTYPE.itemsize == 8
    assert capi.DIGT_CALL_DTYPE.itemsize == 144
    assert capi.SOMATIC_CALL_DTYPE.itemsize == 272


def test_no_cpu_fallback(built):
    """Without a GPU every compute entry point must refuse, with a message -- never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.StrelkaAmdError) as e:
        capi.init(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)
    from strelka_amd import synth
    pb = synth.pileups(4, np.random.default_rng(0))
    with pytest.raises(capi.StrelkaAmdError):
        capi.dependent_eprob(pb)


def test_product_never_touches_oracle():
    """The package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "strelka_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in txt and "strelka_oracle" not in txt and "libstrelka_ref" not in txt, f


def test_integration_doc_lists_the_hooks_the_build_applies():
    """INTEGRATION.md's table of edits is generated from adapter/apply_hooks.py's HOOKS (the list the build applies): the document
    cannot describe a binding that no longer exists"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("apply_hooks", os.path.join(root, "adapter", "apply_hooks.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    assert m.doc_section() in text, "run `python adapter/apply_hooks.py --doc INTEGRATION.md`"
    # every adapter source the document names exists, and every adapter source is named
    import glob
    import re
    named = set(re.findall(r"sk_adapter_[a-z_]+\.(?:cpp|hh)", text))
    present = {os.path.basename(p) for p in glob.glob(os.path.join(root, "adapter", "sk_adapter*"))}
    assert named <= present, sorted(named - present)
    assert {p for p in present if p.endswith(".cpp")} <= named, sorted(present - named)
