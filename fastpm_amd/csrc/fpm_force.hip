// fpm_force.hip -- fastpm_solver_compute_force on one MI355X (reference libfastpm/gravity.c:458-529):
// paint -> normalise -> r2c -> softening -> 3 x (transfer -> c2r) -> readout [-> potential].
// With one rank there are no ghosts (pmghosts.c:67: rank == ThisTask always) and no collectives.
#include "fpm_internal.h"

using namespace fpm;

extern "C" {

int fpmhip_force(fpmhip_plan *p, const fpmhip_particles *pt, int kernel, int softening, double total_mass,
                 void *delta_k_out)
{
    return fpmhip_force_species(p, pt, 1, kernel, softening, total_mass, delta_k_out);
}

static int force_species_body(fpmhip_plan *p, const fpmhip_particles *sets, int nsets, int kernel, int softening,
                              double total_mass, void *delta_k_out);

// A force call is ~30 launches; on the small meshes (configs[0]: 256^3, a kernel every 10 - 20 us) the gaps between them are a
// third of the call.  In the steady state -- the binning has a layout for this many particles, nothing waits on the host
// inside the call -- the call's launches are CAPTURED (on a stream of the plan's own: the caller's may be the null stream,
// which cannot capture), the executable graph of the previous call is updated with them (same topology, new arguments:
// the binning's buffers alternate, the caller's pointers may change) and launched on the caller's stream.  The host
// side of the call runs exactly as without the graph -- nothing about the plan's state is replayed from memory.
// MEASURED (tools/graph_ab.py, profiles/r05_graph_ab.jsonl; ms per force, stream launches -> graph): 128^3 fp32 0.139 -> 0.203,
// 192^3 0.271 -> 0.327, 256^3 fp32 0.415 -> 0.485, 256^3 fp64 0.620 -> 0.692, 384^3 1.46 -> 1.52, 512^3 fp64 4.36 -> 4.44: the
// graph LOSES 0.06 - 0.07 ms per call at every size.  Nothing waits on the host inside a force call, so the host runs
// ahead of the GPU and the stream is never starved -- there are no gaps for a graph to close -- and capture + update +
// launch cost the host more than ~30 plain launches.  Kept as an opt-in A/B: FPMHIP_GRAPH=1.
static bool graph_wanted(const fpmhip_plan *p)
{
    static const int env = getenv("FPMHIP_GRAPH") ? atoi(getenv("FPMHIP_GRAPH")) : 0;
    (void) p;
    return env != 0;
}

// The species loop of gravity.c:279-287 (ghosts), :323-338 (paint every species into one canvas,
// total mass over all of them) and :387-395 (read every species out of each force mesh).
int fpmhip_force_species(fpmhip_plan *p, const fpmhip_particles *sets, int nsets, int kernel, int softening,
                         double total_mass, void *delta_k_out)
{
    if (!p || !sets || nsets < 1) FPM_FAIL(-1, "null argument");
    bool graph = graph_wanted(p) && p->lay.nranks == 1 && p->own_fft && !p->timing && !p->stage_hook && !p->check_hook
                 && !p->capturing && nsets <= 6 && p->geom.paint_mode != FPMHIP_PAINT_ATOMIC;
    for (int si = 0; si < nsets && graph; si++) {
        // a total mass that needs a read-back, an empty set, or a set the binning has no layout for: not the steady state
        if ((total_mass < 0 && sets[si].mass) || sets[si].np <= 0) graph = false;
    }
    if (graph && !(p->layout_np == sets[nsets - 1].np || (p->prebinned && p->binned_np == sets[nsets - 1].np))) graph = false;
    if (!graph) return force_species_body(p, sets, nsets, kernel, softening, total_mass, delta_k_out);

    (void) hipSetDevice(p->device);
    FPM_TRY(check_deferred(p, true));                       // the flags of the previous binning: a host wait, outside the capture
    if (p->layout_np != sets[nsets - 1].np && !p->prebinned)      // (a reported failure resets the layout)
        return force_species_body(p, sets, nsets, kernel, softening, total_mass, delta_k_out);
    // every buffer the call may allocate on the way exists before the capture begins
    FPM_TRY(ensure_buffer(p, BUF_CANVAS)); FPM_TRY(ensure_buffer(p, BUF_F1)); FPM_TRY(ensure_buffer(p, BUF_F2));
    FPM_TRY(ensure_buffer(p, BUF_F0)); FPM_TRY(ensure_buffer(p, BUF_DELTA_K));
    if (!p->cap_stream) FPM_CHECK_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
    hipStream_t user = p->stream;
    hipGraph_t g = nullptr;
    p->stream = p->cap_stream;
    p->capturing = true;
    p->flags_record_deferred = false;
    hipError_t e = hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed);
    int rc = e == hipSuccess ? force_species_body(p, sets, nsets, kernel, softening, total_mass, delta_k_out) : -1;
    hipError_t e2 = e == hipSuccess ? hipStreamEndCapture(p->cap_stream, &g) : e;
    p->capturing = false;
    p->stream = user;
    if (e != hipSuccess || e2 != hipSuccess || !g) {
        if (g) (void) hipGraphDestroy(g);
        p->flags_pending = false;
        if (rc) return rc;
        FPM_FAIL(-1, "hipGraph capture of the force call failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    if (rc) {                                               // the body refused (arguments): nothing has run
        (void) hipGraphDestroy(g);
        p->flags_pending = false;
        return rc;
    }
    bool ready = false;
    if (p->gexec) {
        hipGraphNode_t bad = nullptr;
        hipGraphExecUpdateResult res;
        ready = hipGraphExecUpdate(p->gexec, g, &bad, &res) == hipSuccess;
        if (!ready) { (void) hipGetLastError(); (void) hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
    }
    if (!ready) {
        e = hipGraphInstantiate(&p->gexec, g, nullptr, nullptr, 0);
        p->graph_rebuilds++;
    }
    if (e == hipSuccess) e = hipGraphLaunch(p->gexec, user);
    (void) hipGraphDestroy(g);
    if (e != hipSuccess) { p->flags_pending = false; FPM_FAIL(-1, "hipGraph launch of the force call failed: %s", hipGetErrorString(e)); }
    p->graph_launches++;
    if (p->flags_record_deferred) {
        p->flags_record_deferred = false;
        FPM_CHECK_HIP(hipEventRecord(p->flags_event, user));
    }
    return 0;
}

static int force_species_body(fpmhip_plan *p, const fpmhip_particles *sets, int nsets, int kernel, int softening,
                              double total_mass, void *delta_k_out)
{
    if (!p || !sets || nsets < 1) FPM_FAIL(-1, "null argument");
    if (nsets > 6) FPM_FAIL(-1, "at most FASTPM_SOLVER_NSPECIES = 6 species");
    const fpmhip_particles *pt = &sets[0];
    if (p->lay.nranks != 1)
        FPM_FAIL(-1, "fpmhip_force is the one-rank path; with nranks > 1 drive the stages around the two exchanges");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));                  // gravity.c:169
    if (softening < 0 || softening > FPMHIP_SOFTENING_GAUSSIAN36) FPM_FAIL(-1, "wrong softening kernel type");
    for (int si = 0; si < nsets; si++)
        if (sets[si].np > 0 && !sets[si].acc) FPM_FAIL(-1, "particles without an acc column");

    // FPMHIP_GRADIENT_REAL: one inverse FFT (the potential) + stencil readout, for gradorder = 1
    const bool real_grad = p->geom.gradient_mode == FPMHIP_GRADIENT_REAL && go == 1;
    FPM_TRY(ensure_buffer(p, BUF_CANVAS));
    if (!real_grad) {
        FPM_TRY(ensure_buffer(p, BUF_F1));
        FPM_TRY(ensure_buffer(p, BUF_F2));
    }
    void *canvas = p->buf[BUF_CANVAS];
    void *delta_k = delta_k_out;
    if (!delta_k) {
        FPM_TRY(ensure_buffer(p, BUF_DELTA_K));
        delta_k = p->buf[BUF_DELTA_K];
    }

    if (total_mass < 0) {                                                                 // gravity.c:330-341
        total_mass = 0;
        for (int si = 0; si < nsets; si++) {
            double m = 0;
            FPM_TRY(fpmhip_total_mass(p, &sets[si], &m));
            total_mass += m;
        }
    }
    // the readouts below reuse the binning the paint of THIS call makes: no staleness check needed in between
    struct Trust { fpmhip_plan *p; ~Trust() { p->bin_trusted = false; } } trust{p};
    p->bin_trusted = true;
    const double mean_mass_per_cell = total_mass / p->lay.Norm;                           // gravity.c:342
    // Without a softening kernel between them, the forward x pass runs straight on into the transfer and the
    // backward x pass(es) (fpmhip_r2c_transfer_fft_x_backward): delta_k is stored once and not read again.
    static const bool nofuse = getenv("FPMHIP_NOFUSE") != nullptr;                        // A/B
    const bool fuse_x = p->own_fft && softening == FPMHIP_SOFTENING_NONE && !nofuse;
    // Strip tiles (fpm_strips.hip): the z passes happen inside the particle kernels.  Forwards when one species paints
    // the canvas: the paint leaves half-spectrum rows in delta_k.  Backwards always: the readout takes the force
    // meshes as their y passes leave them.
    const bool strips = p->mg.strips != 0 && !real_grad;
    const bool strips_fwd = strips && fuse_x && nsets == 1;
    if (strips_fwd) {
        if (sets[0].np > 0 && !sets[0].x) FPM_FAIL(-1, "particles without positions");
        FPM_TRY(paint_strips(p, pt, 1.0 / mean_mass_per_cell, delta_k, 0, true));
    } else if (nsets == 1) {
        FPM_TRY(fpmhip_paint(p, pt, 1.0 / mean_mass_per_cell, canvas));                   // gravity.c:336-345
    } else {
        // several species: every species' sums go into the canvas unscaled and the canvas is scaled ONCE, the
        // reference's arithmetic (gravity.c:326-338 paint, :342-345 one fastpm_apply_multiply_transfer) -- one rounding
        // of the mesh dtype per species add and one for the scale, not two per species
        FPM_TRY(fpmhip_paint(p, pt, 1.0, canvas));
        for (int si = 1; si < nsets; si++) FPM_TRY(fpmhip_paint_add(p, &sets[si], 1.0, canvas));
        FPM_TRY(fpmhip_mesh_scale(p, canvas, 1.0 / mean_mass_per_cell));
    }
    FPM_TRY(fpmhip_check_point(p, strips_fwd ? delta_k : canvas, "After painting"));          // gravity.c:350
    // x passes around the transfer: from the canvas (z, y passes first) or from the paint's half-spectrum rows
    auto fwd_x = [&](int mode, void *o0, void *o1, void *o2) -> int {
        if (strips_fwd) return strips_y_xfwd_xback(p, delta_k, kernel, mode, o0, o1, o2);
        return fpmhip_r2c_transfer_fft_x_backward(p, canvas, delta_k, kernel, mode, o0, o1, o2);
    };
    if (!fuse_x) {
        FPM_TRY(fpmhip_r2c(p, canvas, delta_k));                                          // gravity.c:351
        FPM_TRY(fpmhip_softening(p, delta_k, softening));                                 // gravity.c:476
    }
    // gravity.c:352 "After r2c", :381 / :383 around every c2r: the fused step holds delta_k only after the fused x pass
    // and the force meshes only right before their readout (strips: before the z pass inside it -- a NaN or an overflow
    // in a row's half spectrum is one in the row)
    auto check_meshes = [&](void *m0, void *m1, void *m2) -> int {
        if (!p->check_hook) return 0;
        FPM_TRY(fpmhip_check_point(p, delta_k, "After r2c"));
        void *m[3] = {m0, m1, m2};
        const char *names[3] = {"After c2r 0", "After c2r 1", "After c2r 2"};
        for (int d = 0; d < 3; d++) if (m[d]) FPM_TRY(fpmhip_check_point(p, m[d], names[d]));
        return 0;
    };

    bool any_pot = false;
    for (int si = 0; si < nsets; si++) any_pot = any_pot || sets[si].potential != nullptr;
    if (real_grad) {
        // the canvas is free again after the out-of-place r2c: it carries the potential
        if (p->own_fft) {
            if (fuse_x) FPM_TRY(fpmhip_r2c_transfer_fft_x_backward(p, canvas, delta_k, kernel, 1, canvas, nullptr, nullptr));
            else FPM_TRY(fpmhip_transfer_fft_x_backward_pot(p, delta_k, canvas, kernel));
            FPM_TRY(fpmhip_fft_yz_backward(p, canvas, canvas));
        } else {
            FPM_TRY(fpmhip_transfer(p, delta_k, canvas, kernel, FPMHIP_FIELD_POTENTIAL));
            FPM_TRY(fpmhip_c2r(p, canvas));
        }
        FPM_TRY(check_meshes(canvas, nullptr, nullptr));
        for (int si = nsets - 1; si >= 0; si--) FPM_TRY(fpmhip_readout_grad(p, &sets[si], canvas, nullptr));
        for (int si = 0; si < nsets; si++)                                                // gravity.c:487-492
            if (sets[si].potential) FPM_TRY(fpmhip_readout1(p, &sets[si], canvas, sets[si].potential, 1, 0));
        return 0;
    }

    // the canvas is free again after the out-of-place r2c: it carries the x component
    void *f[3] = {canvas, p->buf[BUF_F1], p->buf[BUF_F2]};
    static const int three = getenv("FPMHIP_XBACK3") ? atoi(getenv("FPMHIP_XBACK3")) : 0;   // A/B
    if (p->geom.gradient_mode == FPMHIP_GRADIENT_XSTENCIL && strips && p->own_fft && go == 1) {
        // FPMHIP_GRADIENT_XSTENCIL (fastpm_hip.h): ONE mesh through the backward x pass -- the potential --; its y pass makes
        // the y and z components and passes the potential on; the x component's rows are the plane stencil of the
        // potential's (fpmhip_xstencil_rows).  One rank: the same 15 sweeps as the default; it exists for the slabs.
        FPM_TRY(ensure_buffer(p, BUF_F0));
        void *phi = p->buf[BUF_F0];
        if (fuse_x) FPM_TRY(fwd_x(1, f[1], nullptr, nullptr));
        else FPM_TRY(fpmhip_transfer_fft_x_backward_pot(p, delta_k, f[1], kernel));
        FPM_TRY(strips_y_backward_grad2(p, f[1], f[1], f[2], phi, go));
        FPM_TRY(fpmhip_xstencil_rows(p, phi, nullptr, f[0]));
        FPM_TRY(check_meshes(f[0], f[1], f[2]));
        for (int si = nsets - 1; si >= 0; si--)
            FPM_TRY(readout_strips_zc2r(p, &sets[si], f[0], f[1], f[2], 3, sets[si].acc, 3, 0));
        for (int si = 0; si < nsets; si++)                                                // gravity.c:487-492
            if (sets[si].potential)
                FPM_TRY(readout_strips_zc2r(p, &sets[si], phi, nullptr, nullptr, 1, sets[si].potential, 1, 0));
        return 0;
    }
    if (p->own_fft && go == 1 && !three) {
        // one sweep over delta_k: the x component and the potential through their x passes; the y and
        // z gradient factors are applied to the potential in its y pass (they do not depend on kx)
        if (fuse_x) FPM_TRY(fwd_x(2, f[0], f[1], nullptr));
        else FPM_TRY(fpmhip_transfer_fft_x_backward_potx(p, delta_k, f[0], f[1], kernel));
        // the potential column (gravity.c:487-492) rides along: its (y, z) passes from the same read
        void *potmesh = nullptr;
        if (any_pot) {
            FPM_TRY(ensure_buffer(p, BUF_F0));
            potmesh = p->buf[BUF_F0];
        }
        if (strips) {
            FPM_TRY(strips_y_backward(p, f[0]));
            FPM_TRY(strips_y_backward_grad2(p, f[1], f[1], f[2], potmesh, go));
            FPM_TRY(check_meshes(f[0], f[1], f[2]));
            for (int si = nsets - 1; si >= 0; si--)
                FPM_TRY(readout_strips_zc2r(p, &sets[si], f[0], f[1], f[2], 3, sets[si].acc, 3, 0));
            for (int si = 0; si < nsets; si++)
                if (sets[si].potential)
                    FPM_TRY(readout_strips_zc2r(p, &sets[si], potmesh, nullptr, nullptr, 1, sets[si].potential, 1, 0));
            return 0;
        }
        FPM_TRY(fpmhip_fft_yz_backward(p, f[0], f[0]));
        FPM_TRY(fpmhip_fft_yz_backward_grad2(p, f[1], f[1], f[2], potmesh, kernel));
        if (any_pot) {
            FPM_TRY(check_meshes(f[0], f[1], f[2]));
            for (int si = nsets - 1; si >= 0; si--) FPM_TRY(fpmhip_readout3(p, &sets[si], f[0], f[1], f[2]));
            for (int si = 0; si < nsets; si++)
                if (sets[si].potential) FPM_TRY(fpmhip_readout1(p, &sets[si], potmesh, sets[si].potential, 1, 0));
            return 0;
        }
    } else if (p->own_fft) {
        // one sweep over delta_k: the three transfers + the x pass of their inverse transforms
        if (fuse_x) FPM_TRY(fwd_x(0, f[0], f[1], f[2]));
        else FPM_TRY(fpmhip_transfer_fft_x_backward3(p, delta_k, f[0], f[1], f[2], kernel));
        if (strips) {
            for (int d = 0; d < 3; d++) FPM_TRY(strips_y_backward(p, f[d]));
            FPM_TRY(check_meshes(f[0], f[1], f[2]));
            for (int si = nsets - 1; si >= 0; si--)
                FPM_TRY(readout_strips_zc2r(p, &sets[si], f[0], f[1], f[2], 3, sets[si].acc, 3, 0));
        } else {
            for (int d = 0; d < 3; d++) FPM_TRY(fpmhip_fft_yz_backward(p, f[d], f[d]));
        }
    } else {
        for (int d = 0; d < 3; d++) {                                                     // gravity.c:373-397
            FPM_TRY(fpmhip_transfer(p, delta_k, f[d], kernel, d));
            FPM_TRY(fpmhip_c2r(p, f[d]));
        }
    }
    // with several species the last paint's binning belongs to the last one: read that one first
    if (!strips) {
        FPM_TRY(check_meshes(f[0], f[1], f[2]));
        for (int si = nsets - 1; si >= 0; si--) FPM_TRY(fpmhip_readout3(p, &sets[si], f[0], f[1], f[2]));
    }
    if (any_pot) {                                                                        // gravity.c:487-492
        FPM_TRY(fpmhip_transfer(p, delta_k, canvas, kernel, FPMHIP_FIELD_POTENTIAL));
        FPM_TRY(fpmhip_c2r(p, canvas));
        for (int si = 0; si < nsets; si++)
            if (sets[si].potential) FPM_TRY(fpmhip_readout1(p, &sets[si], canvas, sets[si].potential, 1, 0));
    }
    return 0;
}

int fpmhip_force_host(fpmhip_plan *p, const fpmhip_particles *ph, int kernel, int softening, void *delta_k_host)
{
    return fpmhip_force_species_host(p, ph, 1, kernel, softening, delta_k_host);
}

// All species of the solver with host-resident columns (gravity.c:279-287, 323-338, 387-395): the sets are staged
// back to back in one device region, go through fpmhip_force_species, and every set's acc (and potential) comes back.
int fpmhip_force_species_host(fpmhip_plan *p, const fpmhip_particles *sets, int nsets, int kernel, int softening,
                              void *delta_k_host)
{
    if (!p || !sets || nsets < 1) FPM_FAIL(-1, "null argument");
    if (nsets > 6) FPM_FAIL(-1, "at most FASTPM_SOLVER_NSPECIES = 6 species");
    int64_t np = 0;
    bool any_mass = false, any_pot = false;
    for (int si = 0; si < nsets; si++) {
        const fpmhip_particles &ph = sets[si];
        if (ph.np < 0) FPM_FAIL(-1, "negative particle count");
        if (ph.np > 0 && (!ph.x || !ph.acc)) FPM_FAIL(-1, "store without x or acc columns");
        np += ph.np;
        any_mass |= ph.mass != nullptr;
        any_pot |= ph.potential != nullptr;
    }
    fpm::HostStage *st = &p->host_stage;      // device staging of the host columns, owned by the plan
    if (np > st->cap || (any_mass && !st->mass) || (any_pot && !st->pot)) {
        if (st->x) { (void) hipFree(st->x); (void) hipFree(st->acc); }
        if (st->mass) (void) hipFree(st->mass);
        if (st->pot) (void) hipFree(st->pot);
        *st = fpm::HostStage();
        int64_t cap = std::max<int64_t>(np + np / 16, 1024);
        FPM_CHECK_HIP(hipMalloc(&st->x, cap * 3 * sizeof(double)));
        FPM_CHECK_HIP(hipMalloc(&st->acc, cap * 3 * sizeof(float)));
        if (any_mass) FPM_CHECK_HIP(hipMalloc(&st->mass, cap * sizeof(float)));
        if (any_pot) FPM_CHECK_HIP(hipMalloc(&st->pot, cap * sizeof(float)));
        st->cap = cap;
    }
    fpmhip_particles pd[6];
    int64_t off = 0;
    for (int si = 0; si < nsets; si++) {
        const fpmhip_particles &ph = sets[si];
        pd[si] = ph;
        pd[si].x = st->x + 3 * off;
        pd[si].mass = ph.mass ? st->mass + off : nullptr;
        pd[si].acc = st->acc + 3 * off;
        pd[si].potential = ph.potential ? st->pot + off : nullptr;
        if (ph.np > 0) {
            FPM_CHECK_HIP(hipMemcpyAsync(st->x + 3 * off, ph.x, ph.np * 3 * sizeof(double), hipMemcpyHostToDevice,
                                         p->stream));
            if (ph.mass)
                FPM_CHECK_HIP(hipMemcpyAsync(st->mass + off, ph.mass, ph.np * sizeof(float), hipMemcpyHostToDevice,
                                             p->stream));
        }
        off += ph.np;
    }
    FPM_TRY(ensure_buffer(p, BUF_DELTA_K));
    // The steady-state binning reports what only the device knows (a slab overflow beyond the arrays, a particle outside
    // the rank's region) AFTER its call returned; this entry point synchronises anyway, so it asks here -- the caller
    // reads acc on the host next, and an invalid result must not leave with rc = 0.  One overflow is repaired in place:
    // the arrays grow (check_deferred left the size) and the step runs again.
    for (int attempt = 0; ; attempt++) {
        FPM_TRY(fpmhip_force_species(p, pd, nsets, kernel, softening, -1.0, p->buf[BUF_DELTA_K]));
        FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
        const int rc = fpm::check_deferred(p, true);
        if (rc == -5 && attempt == 0) continue;
        if (rc != 0) return rc;
        break;
    }
    for (int si = 0; si < nsets; si++) {
        const fpmhip_particles &ph = sets[si];
        if (ph.np == 0) continue;
        FPM_CHECK_HIP(hipMemcpyAsync(ph.acc, pd[si].acc, ph.np * 3 * sizeof(float), hipMemcpyDeviceToHost, p->stream));
        if (ph.potential)
            FPM_CHECK_HIP(hipMemcpyAsync(ph.potential, pd[si].potential, ph.np * sizeof(float), hipMemcpyDeviceToHost,
                                         p->stream));
    }
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));   // the caller reads acc on the host right away
    if (delta_k_host) FPM_TRY(fpmhip_export_delta_k(p, p->buf[BUF_DELTA_K], delta_k_host));
    return 0;
}

}  // extern "C"
