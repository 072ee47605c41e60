"""The drop-in boundary, type-checked against the REFERENCE's own headers.

fastpm_amd/host/gravity_hip.c is the translation unit a libfastpm maintainer lists instead of gravity.o: it defines
the three public symbols of api/fastpm/gravity.h:5-22 with the reference's signatures and reads the reference's
structs (FastPMSolver, PM, FastPMPainter, FastPMStore, FastPMCosmology, FastPMFieldDescr, FastPMClock).  The
reference cannot be built in this image (GSL, PFFT absent), but its headers are there: `gcc -fsyntax-only` with the
reference's include paths checks every prototype and every member access.  GSL appears in those headers only as
pointer members (api/fastpm/FDinterp.h:1-8), so three opaque typedefs written here stand in for <gsl/gsl_spline.h>;
MPI is the image's MPICH.  Build-container only: skipped where /root/reference does not exist (the GPU box), and no
reference file is copied anywhere."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
MPI_INC = os.path.join(os.environ.get("FPM_MPI_ROOT", "/opt/conda"), "include")

needs_reference = pytest.mark.skipif(
    not (os.path.isdir(os.path.join(REF, "api", "fastpm")) and os.path.exists(os.path.join(MPI_INC, "mpi.h"))),
    reason="needs the reference's headers and an mpi.h (build container only)")


def _syntax_only(tmp_path, source, extra=()):
    gsl = tmp_path / "gsl"
    gsl.mkdir(exist_ok=True)
    (gsl / "gsl_spline.h").write_text(                         # type-only stand-in, see the module docstring
        "typedef struct gsl_interp gsl_interp;\ntypedef struct gsl_interp_accel gsl_interp_accel;\n"
        "typedef struct gsl_spline gsl_spline;\n")
    cmd = ["gcc", "-std=gnu99", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(REF, "api"),
           "-I" + os.path.join(REF, "libfastpm"), "-I" + str(tmp_path), "-I" + MPI_INC,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "fastpm_amd", "host"), *extra, source]
    return subprocess.run(cmd, capture_output=True, text=True)


@needs_reference
@pytest.mark.parametrize("tu", ["gravity_hip.c", "factors_hip.c", "store_hip.c", "transfer_hip.c", "pm2lpt_hip.c"])
def test_the_binding_type_checks_against_the_reference_headers(tmp_path, tu):
    """gravity_hip.c (the force) and, round 4, the resident drop-in beside it: factors_hip.c (fastpm_kick_store /
    fastpm_drift_store), store_hip.c (fastpm_store_wrap / _decompose / _summary + the sync calls), transfer_hip.c
    (fastpm_apply_decic_transfer / fastpm_powerspectrum_init_from_delta)."""
    r = _syntax_only(tmp_path, os.path.join(ROOT, "fastpm_amd", "host", tu))
    assert r.returncode == 0, r.stderr[-3000:]


@needs_reference
def test_the_resident_symbols_have_the_reference_signatures(tmp_path):
    """Prototypes restated from OUR reading of factors.c:175-197, 373-392, store.c:446-475, 485-488, 807-812,
    transfer.c:77-78 and powerspectrum.c:35: a mismatch with the reference's headers is "conflicting types"; and the
    members the new translation units read."""
    probe = tmp_path / "probe2.c"
    probe.write_text('''
#include <mpi.h>
#include <fastpm/libfastpm.h>
#include <fastpm/logging.h>
#include <fastpm/transfer.h>
#include "pmpfft.h"
void fastpm_kick_store(FastPMKickFactor * kick, FastPMStore * pi, FastPMStore * po, double af);
void fastpm_drift_store(FastPMDriftFactor * drift, FastPMStore * pi, FastPMStore * po, double af);
void fastpm_store_wrap(FastPMStore * p, double BoxSize[3]);
int fastpm_store_decompose(FastPMStore * p, fastpm_store_target_func target_func, void * data, MPI_Comm comm);
void fastpm_store_summary(FastPMStore * p, FastPMColumnTags attribute, MPI_Comm comm, const char * fmt, ...);
void fastpm_apply_decic_transfer(PM * pm, FastPMFloat * from, FastPMFloat * to);
void fastpm_powerspectrum_init_from_delta(FastPMPowerSpectrum * ps, PM * pm, const FastPMFloat * delta1_k, const FastPMFloat * delta2_k);
static void members(FastPMKickFactor * k, FastPMDriftFactor * d, FastPMStore * p, FastPMPowerSpectrum * ps) {
    (void) k->forcemode; (void) k->ai; (void) k->af; (void) k->nsamples; (void) k->q1; (void) k->q2;
    (void) k->dda[31]; (void) k->Dv1[31]; (void) k->Dv2[31];
    (void) d->forcemode; (void) d->ai; (void) d->af; (void) d->nsamples; (void) d->Dv1; (void) d->Dv2;
    (void) d->dyyy[31]; (void) d->da1[31]; (void) d->da2[31];
    (void) p->v[0][0]; (void) p->dx1[0][0]; (void) p->dx2[0][0]; (void) p->pgdc; (void) p->meta.a_x; (void) p->meta.a_v;
    (void) p->columns[31]; (void) p->_column_info[31].attribute; (void) p->_column_info[0].dtype[0];
    (void) p->_column_info[0].nmemb; (void) p->attributes; (void) p->np_upper;
    (void) ps->base.k[0]; (void) ps->base.f[0]; (void) ps->base.size; (void) ps->Nmodes[0]; (void) ps->edges[0];
    (void) ps->Volume; (void) ps->k0; (void) ps->pm->Comm2D;
    (void) fastpm_store_find_column_id(p, COLUMN_ACC);
}
int main(void) { (void) members; return FASTPM_FORCE_COLA == 2 && FASTPM_FORCE_ZA == 4 ? 0 : 1; }
''')
    r = _syntax_only(tmp_path, str(probe), extra=("-Wno-unused-function",))
    assert r.returncode == 0, r.stderr[-3000:]


@needs_reference
def test_a_wrong_member_in_the_resident_files_would_be_caught(tmp_path):
    src = open(os.path.join(ROOT, "fastpm_amd", "host", "factors_hip.c")).read()
    assert "pi->meta.a_v" in src
    bad = tmp_path / "factors_bad.c"
    bad.write_text(src.replace("pi->meta.a_v", "pi->meta.a_vel"))
    r = _syntax_only(tmp_path, str(bad))
    assert r.returncode != 0 and "a_vel" in r.stderr


@needs_reference
def test_the_three_public_symbols_have_the_reference_signatures(tmp_path):
    """A second definition with the prototypes copied from OUR reading of gravity.h must be compatible with the
    reference's declarations: conflicting types are a compile error."""
    probe = tmp_path / "probe.c"
    probe.write_text('''
#include <mpi.h>
#include <fastpm/libfastpm.h>
#include <fastpm/logging.h>
#include "pmpfft.h"
/* api/fastpm/gravity.h:5-22, restated: a mismatch with the header included above is "conflicting types" */
void fastpm_kernel_type_get_orders(FastPMKernelType type, int *potorder, int *gradorder, int *difforder, int *deconvolveorder);
void fastpm_solver_compute_force(FastPMSolver * fastpm, PM * pm, FastPMPainter * painter, FastPMSofteningType dealias,
                                 FastPMKernelType kernel, FastPMFloat * delta_k, double Time);
void gravity_apply_kernel_transfer(FastPMKernelType kernel, PM * pm, FastPMFloat * delta_k, FastPMFloat * canvas,
                                   FastPMFieldDescr field);
/* the members gravity_hip.c reads */
static void members(FastPMSolver * s, PM * pm, FastPMPainter * pa, FastPMStore * st) {
    (void) s->cosmology->ncdm_linearresponse; (void) pm->Nproc[1]; (void) pm->Comm2D; (void) pm->NTask;
    (void) pm->ThisTask; (void) pm->Nmesh[0]; (void) pm->BoxSize[0]; (void) pa->support; (void) st->x[0][0];
    (void) st->acc[0][0]; (void) st->mass; (void) st->potential; (void) st->meta.M0; (void) st->np;
    (void) fastpm_solver_get_species(s, FASTPM_SPECIES_CDM);
    (void) pm->IRegion.start[1]; (void) pm->ORegion.size[2]; (void) st->name;
}
/* the log side effects of gravity.c:398-417 and pmapi.c:335-356 that gravity_hip.c reproduces: the functions it calls
   for them, with the argument lists it uses */
static void side_effects(FastPMStore * st, PM * pm) {
    double a[3], b[3], c[3], d[3];
    fastpm_store_summary(st, COLUMN_ACC, pm_comm(pm), "<s->", a, b, c, d);
    fastpm_info("p%s    acc[%d]: %g %g %g %g\\n", st->name, 0, a[0], b[0], c[0], d[0]);
    fastpm_ilog(INFO, "%s: Task %d has %td field values that are out of bounds\\n", "After r2c", pm->ThisTask, (ptrdiff_t) 0);
}
int main(void) { (void) members; (void) side_effects; return FASTPM_SOLVER_NSPECIES == 6 ? 0 : 1; }
''')
    r = _syntax_only(tmp_path, str(probe), extra=("-Wno-unused-function",))
    assert r.returncode == 0, r.stderr[-3000:]


@needs_reference
def test_a_wrong_member_would_be_caught(tmp_path):
    """The check has teeth: the same file with one member renamed does not compile."""
    src = open(os.path.join(ROOT, "fastpm_amd", "host", "gravity_hip.c")).read()
    assert "pm->Nproc[1]" in src
    bad = tmp_path / "gravity_bad.c"
    bad.write_text(src.replace("pm->Nproc[1]", "pm->NprocY"))
    r = _syntax_only(tmp_path, str(bad))
    assert r.returncode != 0 and "NprocY" in r.stderr


BINDING_TUS = ["gravity_hip.c", "factors_hip.c", "store_hip.c", "transfer_hip.c", "pm2lpt_hip.c"]
LIBC = {"atoi", "free", "getenv", "malloc", "calloc", "realloc", "memcmp", "memset", "memcpy", "strchr", "strcmp", "strlen",
        "pow", "sqrt", "fabs", "floor", "isfinite", "__stack_chk_fail", "_GLOBAL_OFFSET_TABLE_"}


def _nm(path, flag):
    out = subprocess.run(["nm", flag, path], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.split()}


@needs_reference
def test_the_binding_objects_close_over_known_libraries(tmp_path):
    """The five translation units cannot be LINKED here (GSL / PFFT absent), but they can be compiled to objects, and an
    object says what it would ask the linker for: every undefined symbol must be (i) an MPI call, (ii) libc / libm,
    (iii) an fpmhip_* export of libfastpm_hip.so, (iv) a fastpm_hip_* symbol one of our host libraries or another of the
    five objects defines, or (v) a name the reference's own headers declare (the `_cpu` names are the reference's
    functions renamed with -D).  A typo in a call, or a helper that exists only in a header,
    would otherwise surface at the maintainer's link line."""
    import glob
    import re
    gsl = tmp_path / "gsl"
    gsl.mkdir(exist_ok=True)
    (gsl / "gsl_spline.h").write_text(
        "typedef struct gsl_interp gsl_interp;\ntypedef struct gsl_interp_accel gsl_interp_accel;\n"
        "typedef struct gsl_spline gsl_spline;\n")
    objs = []
    for tu in BINDING_TUS:
        o = str(tmp_path / (tu[:-2] + ".o"))
        r = subprocess.run(["gcc", "-std=gnu99", "-c", "-fPIC", "-Wall", "-Werror", "-I" + os.path.join(REF, "api"),
                            "-I" + os.path.join(REF, "libfastpm"), "-I" + str(tmp_path), "-I" + MPI_INC,
                            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "fastpm_amd", "host"),
                            os.path.join(ROOT, "fastpm_amd", "host", tu), "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        objs.append(o)
    defined = set().union(*(_nm(o, "--defined-only") for o in objs))
    ours = set()
    for lib in ("libfastpm_hip.so", "libfastpm_hip_host.so", "libfastpm_hip_mpi.so"):
        path = os.path.join(ROOT, "fastpm_amd", lib)
        if not os.path.exists(path):
            pytest.skip(lib + " is not built")
        ours |= {s for s in _nm(path, "-D") if s.startswith(("fpmhip_", "fastpm_hip_"))}
    ref_text = "".join(open(f, errors="replace").read() for f in glob.glob(os.path.join(REF, "api", "fastpm", "*.h"))
                       + glob.glob(os.path.join(REF, "libfastpm", "*.h")))
    ref_names = set(re.findall(r"[A-Za-z_]\w*", ref_text))
    for o in objs:
        for sym in sorted(_nm(o, "-u")):
            base = sym[:-4] if sym.endswith("_cpu") else sym
            ok = (sym.startswith("MPI_") or sym in LIBC or sym in ours or sym in defined
                  or (not sym.startswith(("fpmhip_", "fastpm_hip_")) and base in ref_names))
            assert ok, "%s needs %s: not MPI, libc, one of our exports, nor declared by the reference's headers" % (os.path.basename(o), sym)
    # the three symbols the force binding exists for are DEFINED by it
    assert {"fastpm_solver_compute_force", "fastpm_kernel_type_get_orders", "gravity_apply_kernel_transfer"} <= defined


def test_integration_md_points_at_the_compiled_binding():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "fastpm_amd/host/gravity_hip.c" in text


def test_the_binding_reproduces_the_log_side_effects():
    """gravity.c:398-417 prints `p%s    acc[%d]: min std mean max` per species and gravity.c:350-383 runs
    pm_check_values: the binding does both (the GPU test of the same lines is in test_gpu_chost.py)."""
    src = open(os.path.join(ROOT, "fastpm_amd", "host", "gravity_hip.c")).read()
    assert 'fastpm_info("p%s    acc[%d]: %g %g %g %g\\n"' in src and "fastpm_store_summary(p, COLUMN_ACC" in src
    assert "fpmhip_set_check_hook" in src and "field values that are out of bounds" in src


def test_the_resident_files_are_named_in_integration_md():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for tu in ("factors_hip.c", "store_hip.c", "transfer_hip.c", "fastpm_hip_store_sync", "-Dfastpm_kick_store=fastpm_kick_store_cpu"):
        assert tu in text, tu
