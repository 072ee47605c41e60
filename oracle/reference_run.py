"""reference_run -- CPU restatement of what the reference's driver does AROUND the force path in its own
regression test tests/lightcone.lua (run by tests/run-test-lightcone.sh, pinned by
tests/run-test-lightcone.check): Gaussian IC with the reference's seeded generator -> remove cosmic variance
-> induce correlation from tests/powerspec.txt -> 2LPT -> K D D F K steps with FASTPM kick/drift factors ->
per-step "D^2(a, 1.0) P(k<...)" log line.

TEST INFRASTRUCTURE ONLY.  Its purpose is to PIN the oracle (and, through tests/test_gpu_reference_log.py,
the GPU library) against the golden numbers the reference's test suite holds: every operator on the force
path (paint, r2c, transfer, c2r, readout) and the "next" rows (kick, drift, 2LPT, de-CIC, P(k)) sits between
the seed and those log lines.  Everything here cites the reference file:line it restates; the RANLXD1
generator (a GSL dependency absent from this image) is restated in ic_oracle.c.

`ops` abstracts who executes the mesh / particle operators: OracleOps (this file, the CPU oracle) or the GPU
adapter in tests/test_gpu_reference_log.py.
"""
import os

import numpy as np
from scipy.integrate import quad

from . import pm_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# tests/run-test-lightcone.check, verbatim numbers (nc = 64, boxsize = 512, seed 100, Omega_m = 0.307494,
# time_step = linspace(0.1, 1, 8), force_mode "fastpm", growth_mode "LCDM", pm_nc_factor 1, kernel 1_4)
CHECK = {
    "sigma8_input": "0.815897",
    "dx1": ("5.36177", "5.36177", "5.36177", "5.36177"),
    "dx2": ("0.455678", "0.44748", "0.453293", "0.45215"),
    # second number of the same lines: sigma(8) of the MEASURED spectrum / D1^2; the reference integrates with
    # GSL QAG at epsrel 1e-4 (powerspectrum.c:268), so only ~4 digits are integrator-independent
    "sigma8_measured": [6.20821, 2.54189, 1.55758, 1.14191, 0.928101, 0.805797, 0.731205, 0.682708],
    "plin": [("0.1", "17305.5"), ("0.228571", "17200.9"), ("0.357143", "17110"), ("0.485714", "17064.7"),
             ("0.614286", "17043.4"), ("0.742857", "17028.1"), ("0.871429", "17014.2"), ("1", "17002.2")],
}


# tests/run-test-lightcone-ODE.check: the same run with growth_mode = "ODE" (tests/lightcone-ODE.lua); dx1, dx2 and
# the first line are the same, the later lines differ from the LCDM ones in the 5th-6th digit -- through nothing but
# f1, f2, D2 in the velocity of the initial condition and in the kick / drift factors.
CHECK_ODE = {
    "plin": [("0.1", "17305.5"), ("0.228571", "17201.1"), ("0.357143", "17110.2"), ("0.485714", "17064.9"),
             ("0.614286", "17043.7"), ("0.742857", "17028.4"), ("0.871429", "17014.5"), ("1", "17002.5")],
}


class Cosmology:
    """libfastpm/cosmology.c for T_cmb = 0, no ncdm, w = -1, flat: HubbleEa :185-201, DHubbleEaDa :226-244,
    D2HubbleEaDa2 :246-265, Omega_m :211-215, growth_mode LCDM :374-388, DGrowthFactorDa :403-427,
    D2GrowthFactorDa2 :429-462."""

    def __init__(self, Omega_m, growth_mode="LCDM"):
        self.Om = Omega_m
        self.OL = 1 - Omega_m                          # :48 with Omega_r = Omega_k = 0
        self._g1 = self._growth_int(1.0)
        self.growth_mode = growth_mode
        self._ode = {}
        if growth_mode == "ODE":
            self._ode1 = self._ode_solve(1.0)

    def _ode_solve(self, a):
        """growth_ode_solve (:300-372): {d1, F1, d2, F2} from a = 0.00625 in matter domination; the reference
        integrates with GSL rkf45 at 1e-8, here scipy at 1e-11."""
        from scipy.integrate import solve_ivp
        if a in self._ode:
            return self._ode[a]
        aini = 0.00625

        def rhs(x, y):
            E, dE = self.E(x), self.dEda(x)
            d = [y[1], -(2. + x / E * dE) * y[1] + 1.5 * self.Omega_m(x) * y[0],
                 y[3], -(2. + x / E * dE) * y[3] + 1.5 * self.Omega_m(x) * (y[2] - y[0] * y[0])]
            return [v / x for v in d]
        y0 = [aini, aini, -3. / 7 * aini * aini, 2 * (-3. / 7 * aini * aini)]
        sol = solve_ivp(rhs, (aini, a), y0, rtol=1e-11, atol=1e-13, method="DOP853")
        self._ode[a] = sol.y[:, -1]
        return self._ode[a]

    def E(self, a):
        return np.sqrt(self.Om / a ** 3 + self.OL)

    def dEda(self, a):
        return 0.5 / self.E(a) * (-3 * self.Om / a ** 4)

    def d2Eda2(self, a):
        E, dE = self.E(a), self.dEda(a)
        return 0.5 / E * (12 * self.Om / a ** 5 - 2 * dE ** 2)

    def Omega_m(self, a):
        return self.Om / a ** 3 / self.E(a) ** 2

    def _growth_int(self, a):                          # :267-299, qag epsrel 1e-9
        f = lambda x: (x / (self.Om + (1 - self.Om - self.OL) * x + self.OL * x ** 3)) ** 1.5
        return self.E(a) * quad(f, 0, a, epsrel=1e-11, epsabs=0)[0]

    def growth(self, a):
        """FastPMGrowthInfo in LCDM mode: D1, f1, D2, f2 (:379-387; note D2 > 0, the 3/7 sits in dx2)."""
        if self.growth_mode == "ODE":                  # :389-397
            y, y1 = self._ode_solve(a), self._ode1
            return {"a": a, "D1": y[0] / y1[0], "f1": y[1] / y[0], "D2": y[2] / y1[2], "f2": y[3] / y[2]}
        Om = self.Omega_m(a)
        D1 = self._growth_int(a) / self._g1
        return {"a": a, "D1": D1, "f1": Om ** (5. / 9), "D2": D1 * D1 * (Om / self.Omega_m(1.0)) ** (-1. / 143),
                "f2": 2 * Om ** (6. / 11)}

    def dDda(self, a):                                 # :412-418 (LCDM), :420-421 (ODE)
        if self.growth_mode == "ODE":
            g = self.growth(a)
            return g["f1"] * g["D1"] / a
        E = self.E(a)
        return self.dEda(a) * self.growth(a)["D1"] / E + E * (a * E) ** -3 / self._g1

    def d2Dda2(self, a):                               # :441-449 (LCDM), :450-458 (ODE)
        if self.growth_mode == "ODE":
            g = self.growth(a)
            ans = -(3. + a / self.E(a) * self.dEda(a)) * g["f1"] + 1.5 * self.Omega_m(a)
            return ans * g["D1"] / (a * a)
        E, dE = self.E(a), self.dEda(a)
        return self.d2Eda2(a) * self.growth(a)["D1"] / E - (dE + 3 / a * E) * (a * E) ** -3 / self._g1

    # libfastpm/factors.c:198-231
    def G_p(self, a):
        return self.growth(a)["D1"]

    def g_p(self, a):
        return self.dDda(a)

    def G_f(self, a):
        return a ** 3 * self.E(a) * self.g_p(a)

    def g_f(self, a):
        E, dD = self.E(a), self.dDda(a)
        return 3 * a * a * E * dD + a ** 3 * self.dEda(a) * dD + a ** 3 * E * self.d2Dda2(a)


def _samples(ai, af, n=32):
    i = np.arange(n)
    return ai * (1.0 * (n - 1 - i) / (n - 1)) + af * (1.0 * i / (n - 1))


def kick_tables(c, ai, ac, af):
    """fastpm_kick_init, FASTPM force mode (factors.c:233-311): (dda, Dv1, Dv2) tables of 32 samples."""
    gi = c.growth(ai)
    Dv1i = gi["D1"] * ai * ai * c.E(ai) * gi["f1"]
    Dv2i = gi["D2"] * ai * ai * c.E(ai) * gi["f2"]
    dda, Dv1, Dv2 = [], [], []
    for ae in _samples(ai, af):
        ge = c.growth(ae)
        dda.append(-1.5 * c.Omega_m(ac) * ac * c.E(ac) * (c.G_f(ae) - c.G_f(ai)) / c.g_f(ac))       # :295-298
        Dv1.append(ge["D1"] * ae * ae * c.E(ae) * ge["f1"] - Dv1i)
        Dv2.append(ge["D2"] * ae * ae * c.E(ae) * ge["f2"] - Dv2i)
    return np.array(dda), np.array(Dv1), np.array(Dv2)


def drift_tables(c, ai, ac, af):
    """fastpm_drift_init, FASTPM force mode (factors.c:324-371): (dyyy, da1, da2)."""
    gi = c.growth(ai)
    dyyy, da1, da2 = [], [], []
    for ae in _samples(ai, af):
        ge = c.growth(ae)
        dyyy.append(1 / (ac ** 3 * c.E(ac)) * (c.G_p(ae) - c.G_p(ai)) / c.g_p(ac))                   # :354-356
        da1.append(ge["D1"] - gi["D1"])
        da2.append(ge["D2"] - gi["D2"])
    return np.array(dyyy), np.array(da1), np.array(da2)


class PowerTable:
    """fastpm_funck_init_from_string + fastpm_funck_eval (powerspectrum.c:347-425): bisection, log-log."""

    def __init__(self, path=os.path.join(GOLDEN, "reference_tests_powerspec.txt")):
        t = np.loadtxt(path)
        self.k, self.f = t[:, 0].copy(), t[:, 1].copy()

    def __call__(self, k):
        k = np.asarray(k, dtype=np.float64)
        out = np.ones_like(k)                                            # :396 k == 0 -> 1
        nz = k != 0
        kk = k[nz]
        r = np.clip(np.searchsorted(self.k, kk, side="right"), 1, len(self.k) - 1)    # while (k < fk->k[m]) r = m
        l = r - 1
        lk = np.log(kk)
        f = ((lk - np.log(self.k[l])) * np.log(self.f[r]) + (np.log(self.k[r]) - lk) * np.log(self.f[l])) \
            / (np.log(self.k[r]) - np.log(self.k[l]))
        out[nz] = np.exp(f)
        return out

    def sigma(self, R):
        """fastpm_powerspectrum_sigma (powerspectrum.c:231-279)."""
        def integrand(k):
            kr = R * k
            if kr < 1e-8:
                return 0.0
            w = 3 * (np.sin(kr) / kr ** 3 - np.cos(kr) / kr ** 2)
            return 4 * np.pi * k * k * w * w * float(self(np.array([k]))[0]) / (2 * np.pi) ** 3
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return np.sqrt(quad(integrand, 0, 500.0 / R, limit=20000, epsrel=1e-8)[0])


def initial_delta_k_xyk(N, BoxSize, seed, power, F=np.float64, remove_variance=True):
    """src/fastpm.c:476-523 for lightcone.lua: fastpm_ic_fill_gaussiank (gadget scheme) ->
    fastpm_ic_remove_variance (initialcondition.c:66-98) -> fastpm_ic_induce_correlation (:55-64, transfer.c:
    188-210).  Returns complex [x][y][kz] in the mesh dtype's complex type, and the white-noise variance
    (pm_compute_variance, the 'Variance of input white noise' log line)."""
    C = np.complex128 if F == np.float64 else np.complex64
    g = np.zeros((N, N, N // 2 + 1, 2))
    O.lib().orc_fill_gaussian_gadget(int(N), int(seed), O._p(g))
    wk = (g[..., 0].astype(F) + 1j * g[..., 1].astype(F)).astype(C)     # stored as FastPMFloat
    if remove_variance:                                                  # src/fastpm.c:479-482, lua default false
        a, b = wk.real.astype(np.float64), wk.imag.astype(np.float64)
        phase = np.arctan2(b, a)
        un = np.cos(phase) + 1j * np.sin(phase)
        un[(a == 0) & (b == 0)] = 0
        un = (un.real.astype(F) + 1j * un.imag.astype(F)).astype(C)
    else:
        un = wk
    kk1 = O.k_tables(N, BoxSize)["kk"].astype(np.float64)
    kk = kk1[:, None, None] + kk1[None, :, None] + kk1[None, None, : N // 2 + 1]
    tr = np.sqrt(power(np.sqrt(kk))) * np.sqrt(1.0 / BoxSize ** 3)
    out = ((un.real.astype(np.float64) * tr).astype(F) + 1j * (un.imag.astype(np.float64) * tr).astype(F)).astype(C)
    return out


def measured_sigma(k, p, R=8.0):
    """fastpm_powerspectrum_sigma (powerspectrum.c:231-279) of a MEASURED spectrum: fastpm_funck_eval on the bin
    table (linear interpolation next to the empty k = 0 bin, log-log elsewhere and beyond the last bin)."""
    k, p = np.asarray(k, dtype=np.float64), np.asarray(p, dtype=np.float64)

    def ev(x):
        if x == 0:
            return 1.0
        l, r = 0, len(k) - 1
        while r - l > 1:
            m = (r + l) // 2
            if x < k[m]:
                r = m
            else:
                l = m
        k1, k2, f1, f2 = k[l], k[r], p[l], p[r]
        if f1 <= 0 or f2 <= 0 or k1 == 0 or k2 == 0:
            return ((x - k1) * f2 + (k2 - x) * f1) / (k2 - k1)
        return np.exp(((np.log(x) - np.log(k1)) * np.log(f2) + (np.log(k2) - np.log(x)) * np.log(f1)) / (np.log(k2) - np.log(k1)))

    def integrand(x):
        kr = R * x
        if kr < 1e-8:
            return 0.0
        w = 3 * (np.sin(kr) / kr ** 3 - np.cos(kr) / kr ** 2)
        return 4 * np.pi * x * x * w * w * ev(x) / (2 * np.pi) ** 3
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return np.sqrt(quad(integrand, 0, 500.0 / R, limit=5000, epsrel=1e-7, points=list(k[1:]))[0])


def column_std(col):
    """fastpm_store_summary(..., "s") (store.c:807-908): sqrt(<x^2> - <x>^2) per member, sums in double."""
    a = np.asarray(col, dtype=np.float64)
    return np.sqrt((a * a).mean(0) - a.mean(0) ** 2)


def large_scale_power(k, p, nmodes, Nmax, k0):
    """fastpm_powerspectrum_large_scale (powerspectrum.c:170-184)."""
    kmax = Nmax * k0
    num = den = 0.0
    i = 0
    while i == 0 or (i < len(k) and k[i] <= kmax):
        num += p[i] * nmodes[i]
        den += nmodes[i]
        i += 1
    return num / den


class OracleOps:
    """The operators of the run, executed by the CPU oracle."""

    def __init__(self, N, BoxSize, precision=64, threads=8):
        self.pm = O.PMOracle(N, BoxSize, precision, threads=threads)
        self.N, self.L = N, BoxSize

    def lpt(self, dk_xyk, q):
        dk = self.pm.alloc()
        self.pm.complex_view(dk)[...] = np.transpose(dk_xyk.astype(self.pm.C), (1, 2, 0))
        return O.pm_2lpt_solve(self.pm, dk, q, shift=(0.0, 0.0, 0.0), kernel=O.KERNELS["1_4"])

    def force(self, x):
        """fastpm_do_force (solver.c:404-478): acc, then P(k) sums of the de-CIC'ed delta_k."""
        r = O.compute_force(self.pm, x)
        d = self.pm.alloc()
        self.pm.decic(r["delta_k"], d)
        return r["acc"], O.powerspectrum_finalize(*self.pm.powerspectrum_sums(d), self.L)

    def kick(self, dda, acc, v):
        return O.kick(O.FORCE_MODES["fastpm"], dda, 0.0, 0.0, 0.0, 0.0, acc, v)

    def drift(self, dyyy, x, v):
        return O.drift(O.FORCE_MODES["fastpm"], dyyy, 0.0, 0.0, 0.0, 0.0, x, v)

    def wrap(self, x):
        return O.store_wrap(x, self.L)


def run_lightcone_test(ops, N=64, BoxSize=512.0, seed=100, Omega_m=0.307494, time_step=None, F=np.float64,
                       nsteps=None, growth_mode="LCDM", lpt_ops=None, remove_variance=True):
    """tests/lightcone.lua up to the quantities the .check file pins.  Returns a dict of log values.
    N = nc (particles per side = the 2LPT mesh, lpt_nc_factor = 1); `ops` executes the force-side operators on ITS
    mesh (nc * pm_nc_factor), `lpt_ops` (default: ops) the initial field and 2LPT on the nc mesh."""
    time_step = np.linspace(0.1, 1.0, 8) if time_step is None else np.asarray(time_step)
    lpt_ops = ops if lpt_ops is None else lpt_ops
    if nsteps is not None:
        time_step = time_step[: nsteps + 1]
    c = Cosmology(Omega_m, growth_mode)
    power = PowerTable()
    log = {"sigma8_input": power.sigma(8.0)}
    if hasattr(lpt_ops, "initial_delta_k"):                             # the operator under test makes the field
        dk = lpt_ops.initial_delta_k(seed, power.k, power.f, **({} if remove_variance else {"remove_variance": False}))
    else:                                                               # itself (src/fastpm.c:476-523)
        dk = initial_delta_k_xyk(N, BoxSize, seed, power, F, remove_variance)
    g = np.arange(N) * (BoxSize / N)                                     # store.c:659-712, shift = 0
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    dx1, dx2 = lpt_ops.lpt(dk, q)
    log["dx1"], log["dx2"] = column_std(dx1), column_std(dx2)
    a0 = float(time_step[0])
    gi = c.growth(a0)                                                    # pm2lpt.c:168-210
    Dv1 = gi["D1"] * a0 * a0 * c.E(a0) * gi["f1"]
    Dv2 = gi["D2"] * a0 * a0 * c.E(a0) * gi["f2"]
    x, v = O.pm_2lpt_evolve(q, np.zeros((len(q), 3), dtype=np.float32), dx1, dx2, gi["D1"], gi["D2"], Dv1, Dv2)
    k0 = 2 * np.pi / BoxSize
    log["plin"], log["sigma8_measured"], log["vstd"] = [], [], []
    a_v = a0

    def force(a):
        nonlocal x
        x = ops.wrap(x)                                                  # fastpm_decompose: store_wrap, solver.c:577
        log["vstd"].append((a_v, column_std(v)))                         # report_domain, src/fastpm.c:1696-1704
        acc, (k, p, nm) = ops.force(x)
        D1 = c.growth(a)["D1"]
        log["plin"].append((a, large_scale_power(k, p, nm, 4, k0) / D1 ** 2))     # src/fastpm.c:1736-1746
        log["sigma8_measured"].append(measured_sigma(k, p) / D1 ** 2)
        return acc

    lookup = lambda tabs, ai, af, a: O.factor_lookup(ai, af, tabs, a)[0]
    acc = force(a0)
    for i in range(len(time_step) - 1):                                  # solver.c:289-296 K D D F K, timemachine.c:68-88
        ai, af = float(time_step[i]), float(time_step[i + 1])
        ac = float(np.exp(0.5 * np.log(ai) + 0.5 * np.log(af)))
        t = kick_tables(c, ai, ai, ac)                                   # kick: a.i = a_v, a.r = force time, a.f
        v = ops.kick(lookup(t, ai, ac, ac) - lookup(t, ai, ac, ai), acc, v)
        t = drift_tables(c, ai, ac, ac)                                  # drift: a.r = a_v
        x = ops.drift(lookup(t, ai, ac, ac) - lookup(t, ai, ac, ai), x, v)
        t = drift_tables(c, ac, ac, af)
        x = ops.drift(lookup(t, ac, af, af) - lookup(t, ac, af, ac), x, v)
        a_v = ac
        acc = force(af)
        t = kick_tables(c, ac, af, af)
        v = ops.kick(lookup(t, ac, af, af) - lookup(t, ac, af, ac), acc, v)
    return log


# tests/run-test-restart.sh:12-13: the two log lines tests/restart.lua must print (nc = 128, boxsize = 384,
# pm_nc_factor = 2 -> 256^3 force mesh, lpt_nc_factor = 1, seed 100, time_step {0.1, 0.5, 0.75, 1}, force_mode
# "fastpm", kernel 1_4, growth_mode left at the lua default "ODE", remove_cosmic_variance left at false)
CHECK_RESTART = {
    "vstd": [("0.6124", ("1.63807", "1.75754", "1.94999")), ("0.8660", ("2.44703", "2.62561", "2.90857"))],
}


def run_restart_test(force_ops, lpt_ops, F=np.float64, nsteps=None):
    """tests/restart.lua up to the 'Velocity dispersion (a = ...)' lines report_domain prints before the forces at
    a = 0.75 and a = 1 (src/fastpm.c:1696-1704): the std of the velocity column after the kicks that used the
    accelerations from the 256^3 mesh at a = 0.1, 0.5 (first line) and 0.75 (second line)."""
    return run_lightcone_test(force_ops, N=128, BoxSize=384.0, seed=100, time_step=[0.1, 0.5, 0.75, 1.0], F=F,
                              nsteps=nsteps, growth_mode="ODE", lpt_ops=lpt_ops, remove_variance=False)


def matches(value, text):
    """Does `value` print as `text` under the reference's "%g" (6 significant digits)?"""
    return ("%g" % value) == text
