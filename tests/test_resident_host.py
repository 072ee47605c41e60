"""The host logic of the resident drop-in (fastpm_amd/host/fastpm_resident_hip.c: which copy of a buffer is the newer one)
run WITHOUT a GPU: the registry's memory back end is replaced by host stand-ins (malloc / memmove behind ctypes
callbacks) that count every copy.  What must hold: a buffer is uploaded once, a device-side write makes the device copy
the newer one, bytes come home only on fastpm_hip_host_sync, a host write (fastpm_hip_host_touched) forces the next
upload, a regrow keeps device-newer data, and a k-space mesh whose host address was reused is noticed by its tag."""
import ctypes

import numpy as np
import pytest

from fastpm_amd import chost


class FakeDevice:
    """"device" memory = ctypes buffers on the host; the k-mesh import / export reverse the byte order of the buffer, so
    that a test can tell a layout-converting copy from a plain one."""

    def __init__(self, kbytes=64):
        self.bufs, self.log, self.kbytes = {}, [], kbytes
        B = chost.MirrorBackend
        self.backend = B(B.ALLOC(self.alloc), B.RELEASE(self.release), B.COPY(self.h2d), B.COPY(self.d2h), B.COPY(self.d2d),
                         B.KCOPY(self.import_k), B.KCOPY(self.export_k),
                         ctypes.CFUNCTYPE(ctypes.c_size_t, ctypes.c_void_p)(lambda plan: self.kbytes))

    def alloc(self, out, n):
        b = ctypes.create_string_buffer(n)
        a = ctypes.addressof(b)
        self.bufs[a] = b
        out[0] = a
        return 0

    def release(self, p):
        del self.bufs[p]
        return 0

    def h2d(self, plan, dst, src, n):
        self.log.append(("h2d", n))
        ctypes.memmove(dst, src, n)
        return 0

    def d2h(self, plan, dst, src, n):
        self.log.append(("d2h", n))
        ctypes.memmove(dst, src, n)
        return 0

    def d2d(self, plan, dst, src, n):
        self.log.append(("d2d", n))
        ctypes.memmove(dst, src, n)
        return 0

    def _rev(self, dst, src):
        raw = ctypes.string_at(src, self.kbytes)
        ctypes.memmove(dst, raw[::-1], self.kbytes)

    def import_k(self, plan, host, dev):
        self.log.append(("import", self.kbytes))
        self._rev(dev, host)
        return 0

    def export_k(self, plan, dev, host):
        self.log.append(("export", self.kbytes))
        self._rev(host, dev)
        return 0


@pytest.fixture
def dev():
    H = chost.host_library()
    d = FakeDevice()
    H.fastpm_hip_mirror_release_all()
    H.fastpm_hip_mirror_set_backend(ctypes.byref(d.backend))
    H.fastpm_hip_mirror_reset_stats()
    yield d
    H.fastpm_hip_mirror_release_all()
    H.fastpm_hip_mirror_set_backend(None)
    assert not d.bufs                      # every twin was freed


PLAN = ctypes.c_void_p(0x1000)             # never dereferenced by the stand-ins


def _dev_array(ptr, like):
    return np.frombuffer(ctypes.string_at(ptr, like.nbytes), dtype=like.dtype).reshape(like.shape)


def test_upload_once_then_resident(dev):
    H = chost.host_library()
    x = np.arange(30, dtype=np.float64)
    p1 = H.fastpm_hip_dev_in(PLAN, x.ctypes.data, x.nbytes)
    p2 = H.fastpm_hip_dev_in(PLAN, x.ctypes.data, x.nbytes)
    assert p1 and p1 == p2 and dev.log == [("h2d", x.nbytes)]
    assert np.array_equal(_dev_array(p1, x), x)
    s = chost.mirror_stats()
    assert (s.h2d_bytes, s.h2d_copies, s.d2h_bytes, s.entries) == (x.nbytes, 1, 0, 1)
    assert not H.fastpm_hip_host_is_stale(x.ctypes.data)
    assert H.fastpm_hip_host_sync(x.ctypes.data) == 0 and len(dev.log) == 1          # nothing to bring home


def test_device_write_sync_and_host_touch(dev):
    H = chost.host_library()
    acc = np.zeros(12, dtype=np.float32)
    p = H.fastpm_hip_dev_out(PLAN, acc.ctypes.data, acc.nbytes)
    assert p and dev.log == []                                   # an output column is never uploaded
    ctypes.memmove(p, np.full(12, 7, dtype=np.float32).ctypes.data, acc.nbytes)     # "the kernel wrote acc"
    assert H.fastpm_hip_host_is_stale(acc.ctypes.data) and acc[0] == 0
    assert H.fastpm_hip_dev_in(PLAN, acc.ctypes.data, acc.nbytes) == p and dev.log == []     # the kick reads it in place
    assert H.fastpm_hip_host_sync(acc.ctypes.data) == 0
    assert dev.log == [("d2h", acc.nbytes)] and (acc == 7).all()
    assert H.fastpm_hip_host_sync(acc.ctypes.data) == 0 and len(dev.log) == 1      # second sync: nothing moves
    # host code rewrites the column (a reader, a permutation): the next device use uploads again
    acc[:] = 3
    H.fastpm_hip_host_touched(acc.ctypes.data)
    p = H.fastpm_hip_dev_inout(PLAN, acc.ctypes.data, acc.nbytes)
    assert dev.log[-1] == ("h2d", acc.nbytes) and (_dev_array(p, acc) == 3).all()
    assert H.fastpm_hip_host_is_stale(acc.ctypes.data)                             # inout: the device copy is newer again


def test_regrow_keeps_device_newer_rows_and_refuses_unknown_rows(dev):
    H = chost.host_library()
    big = np.zeros(4000, dtype=np.float32)
    p = H.fastpm_hip_dev_out(PLAN, big.ctypes.data, 400)                           # np = 100 rows of 4 bytes
    ctypes.memmove(p, np.arange(100, dtype=np.float32).ctypes.data, 400)
    # more rows than the device-newer copy covers (np grew on the host without a sync): refused, not guessed
    assert not H.fastpm_hip_dev_in(PLAN, big.ctypes.data, 800)
    assert b"fewer bytes" in H.fastpm_hip_mirror_error()
    # a device-side overwrite of more rows regrows the allocation and keeps what was there
    p2 = H.fastpm_hip_dev_out(PLAN, big.ctypes.data, 16000)
    assert p2 and ("d2d", 400) in dev.log
    assert np.array_equal(_dev_array(p2, np.zeros(100, dtype=np.float32)), np.arange(100, dtype=np.float32))
    assert H.fastpm_hip_host_sync(big.ctypes.data) == 0 and dev.log[-1] == ("d2h", 16000)


def test_kmesh_goes_through_the_layout_conversion_and_notices_address_reuse(dev):
    H = chost.host_library()
    mesh = np.arange(8, dtype=np.float64)                          # kbytes = 64
    d = H.fastpm_hip_kmesh_in(PLAN, mesh.ctypes.data)
    assert dev.log == [("import", 64)]
    assert ctypes.string_at(d, 64) == mesh.tobytes()[::-1]
    # the force leaves delta_k on the device: the host buffer is tagged (a NaN a host reader would trip over)
    d = H.fastpm_hip_kmesh_out(PLAN, mesh.ctypes.data)
    assert H.fastpm_hip_host_is_stale(mesh.ctypes.data) and np.isnan(mesh[0]) and np.isnan(mesh[1]) and mesh[2] == 2
    ctypes.memmove(d, bytes(range(64)), 64)
    # de-CIC in place, P(k): the twin is used, nothing moves
    assert H.fastpm_hip_kmesh_inout(PLAN, mesh.ctypes.data) == d and H.fastpm_hip_kmesh_in(PLAN, mesh.ctypes.data) == d
    assert dev.log == [("import", 64)]
    assert H.fastpm_hip_host_sync(mesh.ctypes.data) == 0 and dev.log[-1] == ("export", 64)
    assert mesh.tobytes() == bytes(range(64))[::-1]
    # next step: device-newer again, then pm_free / pm_alloc hand the address to someone who clears it (pmapi.c:14)
    H.fastpm_hip_kmesh_out(PLAN, mesh.ctypes.data)
    mesh[:] = 0
    assert not H.fastpm_hip_host_is_stale(mesh.ctypes.data)        # the tag is gone: the host copy is the live one
    n = len(dev.log)
    H.fastpm_hip_kmesh_in(PLAN, mesh.ctypes.data)
    assert dev.log[n:] == [("import", 64)]


def test_release_and_stats(dev):
    H = chost.host_library()
    a, b = np.zeros(10), np.zeros(20)
    H.fastpm_hip_dev_in(PLAN, a.ctypes.data, a.nbytes)
    H.fastpm_hip_dev_in(PLAN, b.ctypes.data, b.nbytes)
    s = chost.mirror_stats()
    assert s.entries == 2 and s.dev_bytes >= a.nbytes + b.nbytes and len(dev.bufs) == 2
    H.fastpm_hip_mirror_release(a.ctypes.data)
    assert chost.mirror_stats().entries == 1 and len(dev.bufs) == 1
    H.fastpm_hip_host_touched(a.ctypes.data)                      # unknown addresses are ignored
    assert H.fastpm_hip_host_sync(a.ctypes.data) == 0


def test_host_library_exports_the_resident_layer():
    H = chost.host_library()
    for name in ("fastpm_solver_compute_force_resident_hip", "fastpm_kick_store_resident_hip",
                 "fastpm_drift_store_resident_hip", "fastpm_store_wrap_resident_hip", "fastpm_store_sync_host_hip",
                 "fastpm_store_host_touched_hip", "fastpm_apply_decic_transfer_resident_hip",
                 "fastpm_powerspectrum_init_from_delta_resident_hip", "fastpm_hip_resident_force",
                 "fastpm_hip_resident_kick", "fastpm_hip_resident_drift", "fastpm_hip_resident_wrap",
                 "fastpm_hip_resident_decic", "fastpm_hip_resident_powerspectrum", "fastpm_hip_resident_summary",
                 "fastpm_hip_lookup3"):
        assert hasattr(H, name), name


def test_mirror_check_notices_a_host_write_behind_a_twin(tmp_path):
    """FASTPM_HIP_MIRROR_CHECK=1 (debug aid): plain-column twins are keyed on the host address alone -- a column
    freed and handed out again at the same address, or written by host code that forgot fastpm_hip_store_touched, would be
    served from the stale device copy.  With the check on, the next use compares the ends of the two copies and fails
    loudly.  The switch is read once per process: a child process with the stand-in back end."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, ctypes, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from fastpm_amd import chost\n"
        "from test_resident_host import FakeDevice, PLAN\n"
        "H = chost.host_library(); d = FakeDevice(); H.fastpm_hip_mirror_set_backend(ctypes.byref(d.backend))\n"
        "x = np.arange(300, dtype=np.float64)\n"
        "assert H.fastpm_hip_dev_in(PLAN, x.ctypes.data, x.nbytes)            # uploaded: SAME\n"
        "assert H.fastpm_hip_dev_in(PLAN, x.ctypes.data, x.nbytes)            # unchanged: passes the check\n"
        "x[-1] = -5.0                                                          # a host write nobody announced\n"
        "p = H.fastpm_hip_dev_in(PLAN, x.ctypes.data, x.nbytes)\n"
        "print('PTR', p, H.fastpm_hip_mirror_error().decode())\n"
        "H.fastpm_hip_host_touched(x.ctypes.data)\n"
        "assert H.fastpm_hip_dev_in(PLAN, x.ctypes.data, x.nbytes)            # announced: uploaded again\n"
        "H.fastpm_hip_mirror_release_all(); H.fastpm_hip_mirror_set_backend(None)\n" % (os.path.dirname(here), here))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FASTPM_HIP_MIRROR_CHECK="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("PTR")][0]
    assert line.split()[1] == "None" and "mirror check" in line
