"""The C-ABI library loads without a GPU and exports every symbol include/svi_hot.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "svi_hot.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    names = set(re.findall(r"\b(svi_[a-z0-9_]+)\s*\(", src))
    names -= {"svi_allreduce_fn"}
    return names


def test_every_declared_symbol_is_exported_and_bound(svi):
    from svi_mapper_amd import _capi
    names = _header_functions()
    assert len(names) >= 45
    lib = C.CDLL(_capi.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, "declared in svi_hot.h but not exported: %s" % missing
    unbound = sorted(names - set(_capi.SIGNATURES))
    assert not unbound, "exported but not bound in _capi.SIGNATURES: %s" % unbound
    extra = sorted(set(_capi.SIGNATURES) - names)
    assert not extra, "bound but not declared in the header: %s" % extra


def test_version_and_status_strings(svi):
    lib = svi.load_library()
    assert lib.svi_version() == 100
    assert lib.svi_status_string(0) == b"ok"
    assert b"device" in lib.svi_status_string(2)
    assert lib.svi_device_count() >= 0


def test_struct_layouts_match_header():
    """ctypes mirrors must have the same size as the C structs (checked by compiling a probe)."""
    import subprocess
    import tempfile
    from svi_mapper_amd import _capi
    probe = r'''
#include <stdio.h>
#include "svi_hot.h"
#include <stddef.h>
int main(void){ printf("%zu %zu %zu %d %zu %zu %zu %zu %zu\n", sizeof(svi_gate), sizeof(svi_ba_options), sizeof(svi_ba_stats), (int)SVI_PH_COUNT,
  sizeof(svi_track_camera), sizeof(svi_track_stereo_params), sizeof(svi_track_record), offsetof(svi_track_record, uv_left),
  offsetof(svi_track_record, s3_count)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        open(c, "w").write(probe)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).split()
    assert int(out[0]) == C.sizeof(_capi.Gate)
    assert int(out[1]) == C.sizeof(_capi.BaOptions)
    assert int(out[2]) == C.sizeof(_capi.BaStats)
    assert int(out[3]) == len(_capi.SVI_PH_NAMES)
    assert int(out[4]) == C.sizeof(_capi.TrackCamera)
    assert int(out[5]) == C.sizeof(_capi.TrackStereoParams)
    import numpy as np
    rec = np.dtype(_capi.TRACK_RECORD_FIELDS)
    assert int(out[6]) == rec.itemsize == _capi.TRACK_RECORD_SIZE
    assert int(out[7]) == rec.fields["uv_left"][1]
    assert int(out[8]) == rec.fields["s3_count"][1]


def test_no_silent_cpu_fallback(svi):
    """Without a device every handle constructor fails loudly; with one, this test is skipped."""
    if svi.load_library().svi_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(svi.SviError) as e:
        svi.HammingMatcher()
    assert e.value.status == 2
    with pytest.raises(svi.SviError) as e:
        svi.BundleAdjuster(1, 1, 0, 0, 0.5)
    assert e.value.status == 2


def test_product_does_not_touch_the_oracle():
    """The product path may not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "svi_mapper_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "orc_" not in txt and "from oracle" not in txt and "import oracle" not in txt, \
                    "%s references the oracle" % os.path.join(dirpath, f)
