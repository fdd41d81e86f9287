"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mvs.h declares, and
refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

import mvs_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mvs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvs_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = mvs_amd.load_library()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libmvs_hip.so does not export %s" % n
    bound = {n for n, _, _ in mvs_amd.ABI}
    assert bound == set(names), (bound ^ set(names))


def test_library_is_hip_only():
    """the product library must not link the oracle"""
    import subprocess
    out = subprocess.check_output(["readelf", "-d", mvs_amd.LIB_PATH]).decode()
    assert "libamdhip64" in out
    assert "oracle" not in out
    syms = subprocess.check_output(["nm", "-D", mvs_amd.LIB_PATH]).decode()
    assert "orc_" not in syms


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_create_fails_loudly_without_gpu():
    with pytest.raises(mvs_amd.MvsError) as e:
        mvs_amd.Context(64, 48)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)
