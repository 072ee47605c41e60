"""The C-ABI library loads on a machine without a GPU and exports exactly the symbols that
include/fastpm_hip.h declares; every compute entry fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "fastpm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fpmhip_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from fastpm_amd import lib
    L = lib.load_library()
    declared = _declared()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), "declared in fastpm_hip.h but not exported: " + name
    assert sorted(lib.SYMBOLS) == declared, set(lib.SYMBOLS) ^ set(declared)
    assert L.fpmhip_version().startswith(b"fastpm_hip")


def test_struct_layouts_match_header_sizes(tmp_path):
    """ctypes mirrors (fastpm_amd/lib.py) against what the C compiler makes of include/fastpm_hip.h: sizes and the
    offsets of the last members."""
    import subprocess
    from fastpm_amd import lib
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fastpm_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(fpmhip_geom), sizeof(fpmhip_particles), sizeof(fpmhip_layout), offsetof(fpmhip_geom, nranks_y), '
                   'offsetof(fpmhip_layout, chunk_b_elems), sizeof(fpmhip_kick_factor), sizeof(fpmhip_drift_factor)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [ctypes.sizeof(lib.Geom), ctypes.sizeof(lib.Particles), ctypes.sizeof(lib.Layout),
                   lib.Geom.nranks_y.offset, lib.Layout.chunk_b_elems.offset, ctypes.sizeof(lib.KickFactor),
                   ctypes.sizeof(lib.DriftFactor)]


def test_kernel_orders_and_error_convention():
    from fastpm_amd import FastPMHipError, fastpm_kernel_type_get_orders
    assert fastpm_kernel_type_get_orders("1_4") == (0, 1, 1, 0)      # lua default, gravity.c:147-152
    assert fastpm_kernel_type_get_orders("gadget") == (0, 1, 1, 2)
    with pytest.raises(FastPMHipError, match="Wrong kernel type"):
        fastpm_kernel_type_get_orders(42)


def test_no_cpu_fallback():
    import torch
    from fastpm_amd import FastPMHipError, PM, lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(FastPMHipError, match="no HIP device"):
        PM(16, 48.0)
    L = lib.load_library()
    g = lib.Geom(16, 48.0, 64, 1, 0, -1, 0, 0, 0, 0, 1)
    plan = ctypes.c_void_p()
    assert L.fpmhip_plan_create(ctypes.byref(g), None, ctypes.byref(plan)) != 0
    assert b"device" in L.fpmhip_last_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fastpm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".c", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pm_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_vpm_find_logic():
    """vpm_find (libfastpm/vpm.c:9-20) without a GPU: stand-in PM objects."""
    from fastpm_amd import VPM
    v = VPM(128, 384.0, [(0.0, 1), (0.25, 2), (0.5, 3)], make_pm=lambda nmesh: nmesh)
    assert [e[2] for e in v.entries] == [128, 256, 384]
    assert v.find(0.1) == 128 and v.find(0.25) == 256 and v.find(0.49) == 256 and v.find(0.5) == 384 and v.find(1.0) == 384
    w = VPM(128, 384.0, [(0.3, 2), (0.6, 3)], make_pm=lambda nmesh: nmesh)
    assert w.find(0.1) == 256                      # "Start with the first pm", vpm.c:17-18
