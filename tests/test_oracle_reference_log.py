"""PIN: the CPU oracle against the golden numbers the reference's own regression test holds.

tests/run-test-lightcone.check (reference) lists log lines that tests/lightcone.lua must print; the numbers
below are copied from it (oracle/reference_run.py CHECK).  Between the random seed and those lines sits every
operator of the force path -- CIC paint, normalisation, r2c, the k-space transfer, three c2r, CIC readout --
and every "next" row -- 2LPT, kick, drift, wrap, de-CIC, the P(k) estimator -- over SEVEN full time steps
to a = 1.  oracle/reference_run.py restates what the reference's driver does around them (GSL's RANLXD1
generator and the gadget-scheme Gaussian field: oracle/ic_oracle.c; LCDM-mode growth; FASTPM kick/drift
factor tables); the mesh and particle arithmetic is the oracle's (oracle/pm_oracle.c).  The input power
spectrum table is the reference's data file tests/powerspec.txt, kept as a fixture.

Every number must print, under the reference's "%g", exactly as the reference printed it."""
import numpy as np
import pytest

from oracle import reference_run as R


@pytest.fixture(scope="module")
def log():
    return R.run_lightcone_test(R.OracleOps(64, 512.0, 64))


def test_input_table_sigma8(log):
    assert len(R.PowerTable().k) == 1769                     # "Found 1769 pairs of values in input spectrum table"
    assert R.matches(log["sigma8_input"], R.CHECK["sigma8_input"])


def test_2lpt_displacement_dispersions(log):
    """"dx1  : 5.36177 5.36177 5.36177 5.36177" and "dx2  : 0.455678 0.44748 0.453293 0.45215" (src/fastpm.c:
    1650-1668): k tables, laplace and diff transfers, 12 c2r + 1 r2c, readout, store summary.  dx2 depends on
    the phases of every mode, i.e. on the restated RANLXD1 stream and the gadget seed-table walk."""
    for d in range(3):
        assert R.matches(log["dx1"][d], R.CHECK["dx1"][d])
        assert R.matches(log["dx2"][d], R.CHECK["dx2"][d])
    assert R.matches(log["dx1"].mean(), R.CHECK["dx1"][3])
    assert R.matches(log["dx2"].mean(), R.CHECK["dx2"][3])


def test_large_scale_power_after_every_force_of_the_run(log):
    """"D^2(a, 1.0) P(k<0.0490625) = ..." at a = 0.1 ... 1 (src/fastpm.c:1736-1746): the force step's delta_k
    (paint, normalise, r2c), de-CIC, the P(k) estimator -- and, from the second line on, the accelerations of
    all previous force calls through the FASTPM kick and drift factors."""
    assert len(log["plin"]) == len(R.CHECK["plin"]) == 8
    for (a, p), (atext, ptext) in zip(log["plin"], R.CHECK["plin"]):
        assert R.matches(a, atext)
        assert R.matches(p, ptext), (a, p, ptext)


def test_growth_mode_ode_variant_of_the_run():
    """tests/run-test-lightcone-ODE.check: the same run with growth_mode = "ODE".  Its eight lines differ from the
    LCDM ones in the 5th-6th digit (17200.9 -> 17201.1, ...) through nothing but f1, f2 and D2 in the initial
    velocities and in the kick / drift factors -- and are reproduced to the printed digit as well: the comparison
    resolves changes of 1e-5 in what the particles were given between two force calls."""
    log = R.run_lightcone_test(R.OracleOps(64, 512.0, 64), growth_mode="ODE")
    assert [t for _, t in R.CHECK_ODE["plin"]] != [t for _, t in R.CHECK["plin"]]
    for (a, p), (atext, ptext) in zip(log["plin"], R.CHECK_ODE["plin"]):
        assert R.matches(p, ptext), (a, p, ptext)


def test_sigma8_of_the_measured_spectra(log):
    """The second number of the same lines, "Sigma8 = ...": sigma(8 Mpc/h) of the MEASURED spectrum over D1^2 --
    every P(k) bin up to the Nyquist frequency enters.  The reference integrates with GSL QAG at epsrel = 1e-4
    (powerspectrum.c:268), so the comparison is to 3e-4, not to the printed digit."""
    for got, ref in zip(log["sigma8_measured"], R.CHECK["sigma8_measured"]):
        assert abs(got / ref - 1) < 3e-4, (got, ref)


def test_restart_lua_velocity_dispersions_pin_the_b2_force():
    """tests/run-test-restart.sh:12-13 -- the reference's SECOND pinned run, at configs[0]'s geometry: restart.lua is
    128^3 particles, `pm_nc_factor = 2` (256^3 force mesh; the 2LPT runs on the 128^3 mesh), seed 100 without
    remove_cosmic_variance, kernel 1_4, fastpm factors with the lua default growth_mode "ODE", steps {0.1, 0.5, 0.75,
    1}.  report_domain (src/fastpm.c:1696-1704) prints the std of the velocity column before every force; the two
    lines the script greps are the velocities after the kicks that applied the B = 2 accelerations of the forces at
    a = 0.1, 0.5 and 0.75 -- a direct pin of acc -> v on a mesh finer than the particle grid, which the lightcone
    check (pm_nc_factor = 1) cannot give.  Must print exactly as the reference printed it."""
    log = R.run_restart_test(R.OracleOps(256, 384.0, 64), R.OracleOps(128, 384.0, 64))
    got = {("%06.4f" % a): s for a, s in log["vstd"]}
    for atext, stext in R.CHECK_RESTART["vstd"]:
        for d in range(3):
            assert R.matches(got[atext][d], stext[d]), (atext, got[atext], stext)


def test_ranlxd1_stream_is_a_uniform_48_bit_stream():
    import ctypes
    from oracle import pm_oracle as O
    s = np.zeros(4096)
    O.lib().orc_ranlxd1_stream(ctypes.c_ulong(100), len(s), O._p(s))
    assert (s >= 0).all() and (s < 1).all()
    assert np.all(s * 2.0 ** 48 == np.floor(s * 2.0 ** 48))          # 48-bit mantissas
    assert abs(s.mean() - 0.5) < 0.02


def test_nbodykit_lua_rsd_factor_pins_the_expansion_rate():
    """tests/run-test-nbodykit.sh:13 of the reference greps 'RSD factor.*1.140331e-02' (libfastpmio/io.c:254-256:
    1 / (100 a E(a)) at the z = 0.5 output of tests/nbodykit.lua, Omega_m = 0.307494) -- the only other line of that test
    that needs no FOF (its 'sigma8 0.815897' is the input-table line pinned above; the two 'Writing N objects.' lines count
    FOF halos, out of scope).  It pins HubbleEa, which every growth factor and every kick / drift table of the runs above is
    built from."""
    cosmo = R.Cosmology(0.307494)
    a = 1.0 / 1.5
    assert "%e" % (1.0 / (100.0 * a * cosmo.E(a))) == "1.140331e-02"
