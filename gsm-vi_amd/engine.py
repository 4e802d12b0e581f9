"""HipEngine: the device-side operations of the GSM/BaM hot path on torch-ROCm storage.

Every numeric operation is one call into libgsmvi_hip.so on the current torch stream; torch only
allocates tensors.  The fit drivers (gsm.py, bam.py) talk to an *engine object* with this
interface, which is how the CPU test-suite exercises their host logic with an oracle-backed
engine (tests/engines.py) while the product default is always this class.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("gsmvi_amd needs an AMD GPU visible to torch (torch.cuda.is_available() is False); "
                           "there is no CPU fallback.")


def _unheld_entry(pool):
    """The first (pinned tensor, numpy view) entry of a host_score pool whose numpy array nobody outside the pool holds, or None.
    ``sys.getrefcount(e[1])`` is 2 for an array referenced by the pool's tuple alone (the tuple + getrefcount's own argument:
    measured on CPython 3.10; round-5 advice -- the threshold stood at 3 and handed out an array that ONE outside holder, a
    recording wrapper or a retained view such as x[:, :k] whose ``.base`` is the array, still referenced)."""
    import sys
    for e in pool:
        if sys.getrefcount(e[1]) <= 2:
            return e
    return None


class HipEngine:
    """One context (workspace + launch heuristics) on one device; grows on demand."""

    name = "hip"
    factor_max_rows = 256      # largest 2B of the factor-form updates (GSMVI_FACTOR_NMAX; the 2B x 2B chain: one workgroup up to 64,
                               # one-workgroup factorisations up to 128, two-level blocked above)
    bam_max_batch = 1024       # B <= 1024 (round 6; 640 before): the B x B chain's last stage (k_bam_post_big, csrc/gsmvi_bam_small.hip) owns
                               # one entry per thread of a 1024-thread workgroup.  The one-workgroup chain covers B <= 128, larger
                               # batches take the blocked multi-workgroup Cholesky and an O(B^2 D) substitution: they work, untuned

    def __init__(self, device=None, max_D=0, max_B=0):
        self.lib = _lib.load_library()
        _require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device if isinstance(device, int) else torch.device(device).index or 0))
        self._ctx = C.c_void_p()
        self._max_D = 0
        self._max_B = 0
        self._tuning = {}
        self._retired = []         # outgrown contexts: kept alive, a captured hipGraph may still launch into their workspace
        self._ctx_captured = False  # has the current context been used under stream capture (a graph may hold its pointers)
        self._stage = {}           # shape -> [pinned staging tensor, event behind its last host -> device copy] (from_host)
        self._hs = {}              # shape -> pinned buffers of the host-callable round trip (host_score)
        self.host_score_profile = None   # a dict here makes host_score accumulate its three phases (benchmarks only)
        if max_D and max_B:
            self._ensure(max_D, max_B)

    # ---- context management -------------------------------------------------------------
    def _ensure(self, D, B):
        if self._ctx and D <= self._max_D and B <= self._max_B:
            return
        newD, newB = max(D, self._max_D), max(B, self._max_B)
        if self._ctx:
            # The outgrown context is RETIRED, not destroyed: kernels enqueued on it may still be running and, worse, a hipGraph
            # captured earlier (GSM.fit's replayed blocks, a user's torch.cuda.graph around engine calls) holds raw pointers
            # into its workspace.  It costs its workspace (~100 MiB at D=4096) until release_retired() / close().
            # (advisor, round 4) A long-lived engine that sweeps growing shapes must not hoard them: a context that was never
            # used while a stream capture was active cannot be referenced by a graph -- it is destroyed behind a device
            # synchronisation; and the workspace grows geometrically (at least 1.5x per regrow in the dimension that grew), so
            # the retired list of an ever-growing sweep stays logarithmic.
            if self._ctx_captured:
                self._retired.append(self._ctx)
            else:
                torch.cuda.synchronize(self.device)
                self.lib.gsmvi_destroy(self._ctx)
            self._ctx = C.c_void_p()
            self._ctx_captured = False
            if newD > self._max_D:
                newD = max(newD, (3 * self._max_D + 1) // 2)
            if newB > self._max_B:
                newB = max(newB, (3 * self._max_B + 1) // 2)
        ctx = C.c_void_p()
        _lib.check("gsmvi_create", self.lib.gsmvi_create(C.byref(ctx), self.device.index, newD, newB))
        self._ctx, self._max_D, self._max_B = ctx, newD, newB
        for k, v in self._tuning.items():          # knobs survive a context regrow
            _lib.check("gsmvi_set_tuning", self.lib.gsmvi_set_tuning(self._ctx, k.encode(), int(v)))
        word = getattr(self, "_reg_word", None)    # ... and so does the regulariser's device source (bam_reg_source): a regrown
        if word is not None:                       # context that silently fell back to the by-value argument would be WRONG
            _lib.check("gsmvi_bam_set_reg_source", self.lib.gsmvi_bam_set_reg_source(self._ctx, C.c_void_p(word.data_ptr())))

    def release_retired(self):
        """Destroy the contexts a regrow left behind.  Only when no captured graph that used them will be replayed again."""
        if self._retired:
            torch.cuda.synchronize(self.device)
            for c in self._retired:
                self.lib.gsmvi_destroy(c)
            self._retired = []

    def close(self):
        self.release_retired()
        if self._ctx:
            torch.cuda.synchronize(self.device)
            self.lib.gsmvi_destroy(self._ctx)
            self._ctx = C.c_void_p()
            self._max_D = self._max_B = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tuning(self, name, value):
        self._tuning[name] = int(value)
        self._ensure(max(self._max_D, 1), max(self._max_B, 1))
        _lib.check("gsmvi_set_tuning", self.lib.gsmvi_set_tuning(self._ctx, name.encode(), int(value)))

    # ---- array helpers ------------------------------------------------------------------
    def asarray(self, x):
        """float64 device tensor with unit inner stride (copies host data to the device)."""
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=torch.float64)
        else:
            t = torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float64)), device=self.device)
        if t.dim() >= 1 and t.stride(-1) != 1:
            t = t.contiguous()
        return t

    def clone(self, x):
        """Owned device copy (the drivers never alias or mutate the caller's arrays)."""
        return self.asarray(x).clone()

    def to_numpy(self, t):
        return t.detach().to("cpu").numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    # ---- the host-callable round trip of the fit loops (gsmvi/gsm_numpy.py:117: vs = self.lp_g(samples) on host arrays) ----
    def to_host(self, t):
        """A device tensor as a FRESH numpy array, through pinned memory: one non-blocking copy on the current stream and
        one stream synchronisation (a pageable ``.cpu()`` stages through the driver's bounce buffer and blocks twice).  The
        array owns its (pinned) storage -- torch's caching host allocator hands the block out again only after the array is
        dropped -- so a callable that keeps its argument (a recording wrapper, say) is not overwritten by the next iteration."""
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t.detach(), non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return h.numpy()

    def from_host(self, a, out=None):
        """A host array as a float64 device tensor (``out`` when given), through a pinned staging buffer owned by the engine:
        a host memcpy into the buffer, then ONE non-blocking copy on the current stream -- no host synchronisation.  The
        buffer of a shape is reused; an event recorded behind its last copy is waited for before it is overwritten (the fit
        loops never get there early: their next device -> host copy is ordered behind it on the same stream)."""
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        a = np.asarray(a)
        key = tuple(a.shape)
        ent = self._stage.get(key)
        if ent is None:
            if len(self._stage) >= 8:                          # (a fit uses one or two shapes; do not hoard pinned memory)
                self._stage.clear()
            ent = self._stage[key] = [torch.empty(a.shape, dtype=torch.float64, pin_memory=True), None]
        buf, ev = ent
        if ev is not None:
            ev.synchronize()
        np.copyto(buf.numpy(), a, casting="unsafe")           # (also the float32 -> float64 conversion of gsm_numpy.py:47)
        out = self.empty(*a.shape) if out is None else out
        assert tuple(out.shape) == key, f"lp_g returned shape {key}, expected {tuple(out.shape)}"
        out.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        ent[1] = ev
        return out

    def host_score(self, lp_g, X, out=None):
        """vs = lp_g(samples) for a HOST callable (gsm_numpy.py:117; examples/example_gsm_numpy.py:24-29): samples to the host,
        the user's numpy code, scores back to the device.  Per call: one non-blocking device -> host copy into a pinned
        buffer, ONE stream synchronisation, the callable, a host memcpy of its result into a pinned staging buffer and one
        non-blocking host -> device copy -- no allocation, no event, no second synchronisation (round 5; before: pageable
        ``.cpu()`` and ``.to(device)``, two blocking copies through the driver's bounce buffers).
        Buffers are cached per shape.  The samples array handed to the callable comes from a small pool and is reused only
        when nobody else holds a reference to it (a callable that records its argument keeps it intact); the score staging
        buffer is overwritten only behind this call's own synchronisation, which orders it behind the previous upload on the
        same stream (a different stream since the last call: a full device synchronisation first)."""
        key = tuple(X.shape)
        hs = self._hs.get(key)
        if hs is None:
            if len(self._hs) >= 8:
                self._hs.clear()
            gp = torch.empty(key, dtype=torch.float64, pin_memory=True)
            hs = self._hs[key] = {"pool": [], "gpin": gp, "gnp": gp.numpy(), "stream": None}
        stream = torch.cuda.current_stream(self.device)
        if hs["stream"] is not None and hs["stream"] != stream.cuda_stream:
            torch.cuda.synchronize(self.device)
        hs["stream"] = stream.cuda_stream
        ent = _unheld_entry(hs["pool"])
        if ent is None:
            xp = torch.empty(key, dtype=torch.float64, pin_memory=True)
            ent = (xp, xp.numpy())
            if len(hs["pool"]) < 4:
                hs["pool"].append(ent)
        prof = self.host_score_profile                        # None, or a dict the call-path benchmark reads (bench.callpath_rates)
        if prof is not None:
            import time
            t0 = time.perf_counter()
        ent[0].copy_(X.detach(), non_blocking=True)
        stream.synchronize()                                  # (polling an event instead was measured: no difference)
        if prof is not None:
            t1 = time.perf_counter()
        g = lp_g(ent[1])
        if prof is not None:
            t2 = time.perf_counter()
        if isinstance(g, torch.Tensor):
            g = g.detach().cpu().numpy()
        g = np.asarray(g)
        out = self.empty(*key) if out is None else out
        assert g.shape == key and tuple(out.shape) == key, f"lp_g returned shape {g.shape}, expected {key}"
        np.copyto(hs["gnp"], g, casting="unsafe")             # (also float32 -> float64: gsm_numpy.py:47 returns float64)
        out.copy_(hs["gpin"], non_blocking=True)
        if prof is not None:
            t3 = time.perf_counter()
            prof["calls"] = prof.get("calls", 0) + 1
            prof["d2h_and_sync_s"] = prof.get("d2h_and_sync_s", 0.0) + (t1 - t0)     # waits for the device work in front of it too
            prof["callable_s"] = prof.get("callable_s", 0.0) + (t2 - t1)
            prof["stage_and_h2d_enqueue_s"] = prof.get("stage_and_h2d_enqueue_s", 0.0) + (t3 - t2)
        return out

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    def eye(self, D):
        return torch.eye(D, dtype=torch.float64, device=self.device)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64, device=self.device)

    def new_flag(self):
        return torch.zeros(1, dtype=torch.int32, device=self.device)

    def read_flag(self, flag):
        return int(flag.item())          # synchronises

    def flag_tensor(self, flag):
        """the tensor a collective moves for a flag (dist.root_potrf): the flag itself"""
        return flag

    def flag_assign(self, flag, t):
        """adopt the received value (nothing to do: the collective wrote into the flag's own storage)"""
        return flag

    def normal_from_host(self, z_host):
        """Upload a host (B,D) array of standard normals (parity mode: numpy MT19937 stream)."""
        return self.from_host(z_host)

    def normal(self, B, D, seed, call=0, out=None, raw=None):
        """(B, D) standard normals from the counter-based device stream (csrc/gsmvi_rng.hip): a pure function of
        (seed, call, element index).  Replaces the z-stream of np.random.multivariate_normal
        (gsmvi/gsm_numpy.py:105,116); ``call`` is the fit iteration."""
        self._ensure(max(self._max_D, 1), max(self._max_B, 1))
        Z = self.empty(B, D) if out is None else out
        assert Z.is_contiguous() and Z.numel() == B * D
        _lib.check("gsmvi_randn_f64", self.lib.gsmvi_randn_f64(
            self._ctx, self._stream(), int(seed) & (2 ** 64 - 1), int(call), B * D, C.c_void_p(Z.data_ptr()),
            C.c_void_p(raw.data_ptr()) if raw is not None else None))
        return Z

    def normal_batch(self, ncalls, B, D, seed, call0=0, out=None, call_in=None, call_out=None):
        """(ncalls, B, D) standard normals: slice c is bit-identical to ``normal(B, D, seed, call0 + c)`` -- one launch for a
        block of fit iterations (the draw stream does not depend on the state).  ``call_in`` / ``call_out``: device int64
        words (1-element tensors); *call_in is added to call0 on the device and *call_out receives *call_in + ncalls, so a
        graph-captured launch advances through the stream on every replay."""
        self._ensure(max(self._max_D, 1), max(self._max_B, 1))
        Z = self.empty(ncalls, B, D) if out is None else out
        assert Z.is_contiguous() and Z.numel() == ncalls * B * D
        _lib.check("gsmvi_randn_batch_f64", self.lib.gsmvi_randn_batch_f64(
            self._ctx, self._stream(), int(seed) & (2 ** 64 - 1), int(call0), int(ncalls), B * D, C.c_void_p(Z.data_ptr()),
            C.c_void_p(call_in.data_ptr()) if call_in is not None else None,
            C.c_void_p(call_out.data_ptr()) if call_out is not None else None))
        return Z

    def _stream(self):
        if not self._ctx_captured and torch.cuda.is_current_stream_capturing():
            self._ctx_captured = True       # this context's workspace is now referenced by a graph: never destroy it on regrow
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _mat(t, name):
        assert isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64, \
            f"{name}: expected a float64 CUDA tensor"
        assert t.dim() == 2 and t.stride(1) == 1, f"{name}: expected a 2-D tensor with unit inner stride"
        return C.c_void_p(t.data_ptr()), int(t.stride(0)) if t.shape[0] > 1 else int(max(t.stride(0), t.shape[1]))

    @staticmethod
    def _vec(t, name):
        assert isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.dim() == 1 \
            and (t.numel() <= 1 or t.stride(0) == 1), f"{name}: expected a contiguous float64 CUDA vector"
        return C.c_void_p(t.data_ptr())

    # ---- the hot path -------------------------------------------------------------------
    def gsm_update(self, X, G, mu0, S0, out=None, general=False):
        """(mu, S) = gsm_update(samples, vs, mu0, S0)   [gsmvi/gsm_numpy.py:27-55].  S0 must be symmetric (a covariance)
        unless ``general=True``, which reads all of S0 (gsmvi_gsm_update_general_f64: the reference's literal
        S0 + mean semantics for any square S0; not a performance path)."""
        assert X.dim() == 2 and G.dim() == 2            # gsm_numpy.py:43-44
        B, D = X.shape
        assert G.shape == (B, D) and mu0.shape == (D,) and S0.shape == (D, D)
        self._ensure(D, B)
        mu, S = (self.empty(D), self.empty(D, D)) if out is None else out
        px, ldx = self._mat(X, "samples")
        pg, ldg = self._mat(G, "vs")
        ps0, lds0 = self._mat(S0, "S0")
        ps, lds = self._mat(S, "S")
        name = "gsmvi_gsm_update_general_f64" if general else "gsmvi_gsm_update_f64"
        _lib.check(name, getattr(self.lib, name)(
            self._ctx, self._stream(), D, B, px, ldx, pg, ldg, self._vec(mu0, "mu0"), ps0, lds0,
            self._vec(mu, "mu"), ps, lds))
        return mu, S

    def record_len(self, D):
        return int(self.lib.gsmvi_gsm_record_len(int(D)))

    def gsm_local_stage(self, X, G, mu0, S0, out=None):
        """Per-sample records [d | e | dmu] for this rank's samples (batch-sharded path;
        gsmvi/gsm_numpy.py:7-18 for each local sample)."""
        Bl, D = X.shape
        self._ensure(D, Bl)
        rec = self.empty(Bl, self.record_len(D)) if out is None else out
        px, ldx = self._mat(X, "samples")
        pg, ldg = self._mat(G, "vs")
        ps0, lds0 = self._mat(S0, "S0")
        pr, ldr = self._mat(rec, "rec")
        _lib.check("gsmvi_gsm_local_stage_f64", self.lib.gsmvi_gsm_local_stage_f64(
            self._ctx, self._stream(), D, Bl, px, ldx, pg, ldg, self._vec(mu0, "mu0"), ps0, lds0, pr, ldr))
        return rec

    def gsm_apply(self, rec, mu0, S0, out=None):
        """Combined rank-2B update from ALL samples' records (gsmvi/gsm_numpy.py:17-23,50-53)."""
        B = rec.shape[0]
        D = mu0.shape[0]
        self._ensure(D, B)
        mu, S = (self.empty(D), self.empty(D, D)) if out is None else out
        pr, ldr = self._mat(rec, "rec")
        ps0, lds0 = self._mat(S0, "S0")
        ps, lds = self._mat(S, "S")
        _lib.check("gsmvi_gsm_apply_f64", self.lib.gsmvi_gsm_apply_f64(
            self._ctx, self._stream(), D, B, pr, ldr, self._vec(mu0, "mu0"), ps0, lds0, self._vec(mu, "mu"),
            ps, lds))
        return mu, S

    # ---- row-block sharded covariance ---------------------------------
    def gsm_rows_stage(self, G, S0_rows, out=None):
        """Columns [row0, row0+nrows) of G S0 from the owned row block of the symmetric S0 (gsm_numpy.py:7)."""
        B, D = G.shape
        nr = S0_rows.shape[0]
        assert S0_rows.shape == (nr, D)
        self._ensure(D, B)
        SGc = self.empty(B, nr) if out is None else out
        pg, ldg = self._mat(G, "vs")
        ps0, lds0 = self._mat(S0_rows, "S0_rows")
        po, ldo = self._mat(SGc, "SGcols")
        _lib.check("gsmvi_gsm_rows_stage_f64", self.lib.gsmvi_gsm_rows_stage_f64(
            self._ctx, self._stream(), D, B, nr, pg, ldg, ps0, lds0, po, ldo))
        return SGc

    def gsm_records(self, X, G, mu0, SG, out=None):
        """Records [d | e | dmu] of all samples from the gathered SG = G S0 (gsm_numpy.py:8-17)."""
        B, D = X.shape
        assert SG.shape == (B, D) and SG.is_contiguous()
        self._ensure(D, B)
        rec = self.empty(B, self.record_len(D)) if out is None else out
        px, ldx = self._mat(X, "samples")
        pg, ldg = self._mat(G, "vs")
        pr, ldr = self._mat(rec, "rec")
        _lib.check("gsmvi_gsm_records_f64", self.lib.gsmvi_gsm_records_f64(
            self._ctx, self._stream(), D, B, px, ldx, pg, ldg, self._vec(mu0, "mu0"), C.c_void_p(SG.data_ptr()),
            pr, ldr))
        return rec

    def gsm_apply_rows(self, rec, mu0, S0_rows, row0, out=None):
        """(mu, S_rows): the rank-2B update restricted to the owned rows, and the full new mean."""
        B = rec.shape[0]
        D = mu0.shape[0]
        nr = S0_rows.shape[0]
        self._ensure(D, B)
        mu, S = (self.empty(D), self.empty(nr, D)) if out is None else out
        pr, ldr = self._mat(rec, "rec")
        ps0, lds0 = self._mat(S0_rows, "S0_rows")
        ps, lds = self._mat(S, "S_rows")
        _lib.check("gsmvi_gsm_apply_rows_f64", self.lib.gsmvi_gsm_apply_rows_f64(
            self._ctx, self._stream(), D, B, int(row0), nr, pr, ldr, self._vec(mu0, "mu0"), ps0, lds0,
            self._vec(mu, "mu"), ps, lds))
        return mu, S

    def gsm_factor_update(self, Z, X, G, mu0, F0, out=None, flag=None, n_reverts=None):
        """Factor-form update: Sigma = F^T F, X = mu0 + Z F0.  Returns (mu, F, flag); flag != 0 means
        the 2B x 2B positive-definite test failed and (mu, F) = (mu0, F0) (revert; counted in n_reverts)."""
        B, D = Z.shape
        self._ensure(D, B)
        mu, F = (self.empty(D), self.empty(D, D)) if out is None else out
        flag = self.new_flag() if flag is None else flag
        pz, ldz = self._mat(Z, "Z")
        px, ldx = self._mat(X, "X")
        pg, ldg = self._mat(G, "G")
        pf0, ldf0 = self._mat(F0, "F0")
        pf, ldf = self._mat(F, "F")
        _lib.check("gsmvi_gsm_factor_update_f64", self.lib.gsmvi_gsm_factor_update_f64(
            self._ctx, self._stream(), D, B, pz, ldz, px, ldx, pg, ldg, self._vec(mu0, "mu0"), pf0, ldf0,
            self._vec(mu, "mu"), pf, ldf, C.c_void_p(flag.data_ptr()),
            C.c_void_p(n_reverts.data_ptr()) if n_reverts is not None else None))
        return mu, F, flag

    def bam_factor_update(self, Z, X, G, mu0, F0, reg, out=None, flag=None, n_reverts=None):
        """Factor-form BaM update: Sigma = F^T F, X = mu0 + Z F0.  Returns (mu, F, flag); F^T F equals the S of
        ``bam_update`` (jitter 0) to round-off; flag != 0: (mu, F) = (mu0, F0) (revert; counted in n_reverts)."""
        B, D = Z.shape
        self._ensure(D, B)
        mu, F = (self.empty(D), self.empty(D, D)) if out is None else out
        flag = self.new_flag() if flag is None else flag
        pz, ldz = self._mat(Z, "Z")
        px, ldx = self._mat(X, "X")
        pg, ldg = self._mat(G, "G")
        pf0, ldf0 = self._mat(F0, "F0")
        pf, ldf = self._mat(F, "F")
        _lib.check("gsmvi_bam_factor_update_f64", self.lib.gsmvi_bam_factor_update_f64(
            self._ctx, self._stream(), D, B, pz, ldz, px, ldx, pg, ldg, self._vec(mu0, "mu0"), pf0, ldf0, float(reg),
            self._vec(mu, "mu"), pf, ldf, C.c_void_p(flag.data_ptr()),
            C.c_void_p(n_reverts.data_ptr()) if n_reverts is not None else None))
        return mu, F, flag

    def gsm_factor_local_stage(self, Z_l, X_l, G_l, mu0, F0, out=None):
        """Records [x - mu0 | u | u F0] of this rank's samples (batch-sharded factor path)."""
        Bl, D = Z_l.shape
        self._ensure(D, Bl)
        rec = self.empty(Bl, self.record_len(D)) if out is None else out
        pz, ldz = self._mat(Z_l, "Z")
        px, ldx = self._mat(X_l, "X")
        pg, ldg = self._mat(G_l, "G")
        pf0, ldf0 = self._mat(F0, "F0")
        pr, ldr = self._mat(rec, "rec")
        _lib.check("gsmvi_gsm_factor_local_stage_f64", self.lib.gsmvi_gsm_factor_local_stage_f64(
            self._ctx, self._stream(), D, Bl, pz, ldz, px, ldx, pg, ldg, self._vec(mu0, "mu0"), pf0, ldf0, pr, ldr))
        return rec

    def gsm_factor_apply(self, Z, rec, mu0, F0, out=None, flag=None, n_reverts=None):
        """(mu, F, flag) from the replicated draws Z and ALL samples' records (batch-sharded factor path)."""
        B, D = Z.shape
        assert rec.shape[0] == B
        self._ensure(D, B)
        mu, F = (self.empty(D), self.empty(D, D)) if out is None else out
        flag = self.new_flag() if flag is None else flag
        pz, ldz = self._mat(Z, "Z")
        pr, ldr = self._mat(rec, "rec")
        pf0, ldf0 = self._mat(F0, "F0")
        pf, ldf = self._mat(F, "F")
        _lib.check("gsmvi_gsm_factor_apply_f64", self.lib.gsmvi_gsm_factor_apply_f64(
            self._ctx, self._stream(), D, B, pz, ldz, pr, ldr, self._vec(mu0, "mu0"), pf0, ldf0,
            self._vec(mu, "mu"), pf, ldf, C.c_void_p(flag.data_ptr()),
            C.c_void_p(n_reverts.data_ptr()) if n_reverts is not None else None))
        return mu, F, flag

    # ---- column-sharded factor form (dist.col_sharded_gsm_factor_update) --------------------------
    def sample_cols(self, Z, mu_cols, Fcols, out=None):
        """X[:, C] = mu[C] + Z F[:, C]: the owned column slice of x = mu + z F (gsm_numpy.py:116) from the owned block of F."""
        B, D = Z.shape
        nc = Fcols.shape[1]
        assert Fcols.shape == (D, nc) and mu_cols.shape == (nc,)
        self._ensure(D, B)
        X = self.empty(B, nc) if out is None else out
        pz, ldz = self._mat(Z, "Z")
        pf, ldf = self._mat(Fcols, "Fcols")
        px, ldx = self._mat(X, "Xcols")
        _lib.check("gsmvi_sample_cols_f64", self.lib.gsmvi_sample_cols_f64(
            self._ctx, self._stream(), D, B, nc, pz, ldz, self._vec(mu_cols, "mu_cols"), pf, ldf, px, ldx))
        return X

    def gsm_factor_w_partial(self, G, col0, Fcols, out=None):
        """The rank's PARTIAL sum of W = G F^T over its owned columns: G[:, C] F[:, C]^T (B x D) -- gsmvi_gsm_rows_stage_f64 on the
        block (the caller all-reduces the partials)."""
        nc = Fcols.shape[1]
        self._ensure(Fcols.shape[0], G.shape[0])         # (the product's output is D wide: the context must be sized for D, not for the block)
        return self.gsm_rows_stage(G[:, col0:col0 + nc], Fcols, out=out)

    def gsm_factor_apply_cols(self, Z, W, X, mu0, F0cols, col0, out=None, flag=None, n_reverts=None):
        """(mu, Fcols, flag): the factor-form update of the OWNED column block from the replicated draws Z, the all-reduced
        W = G F^T and the gathered samples X; mu is full length, entries C written (gsmvi_gsm_factor_apply_cols_f64)."""
        B, D = Z.shape
        nc = F0cols.shape[1]
        assert W.shape == (B, D) and W.is_contiguous() and X.shape == (B, D) and F0cols.shape == (D, nc)
        self._ensure(D, B)
        mu, F = (self.empty(D), self.empty(D, nc)) if out is None else out
        flag = self.new_flag() if flag is None else flag
        pz, ldz = self._mat(Z, "Z")
        px, ldx = self._mat(X, "X")
        pf0, ldf0 = self._mat(F0cols, "F0cols")
        pf, ldf = self._mat(F, "Fcols")
        _lib.check("gsmvi_gsm_factor_apply_cols_f64", self.lib.gsmvi_gsm_factor_apply_cols_f64(
            self._ctx, self._stream(), D, B, int(col0), nc, pz, ldz, C.c_void_p(W.data_ptr()), px, ldx, self._vec(mu0, "mu0"),
            pf0, ldf0, self._vec(mu, "mu"), pf, ldf, C.c_void_p(flag.data_ptr()),
            C.c_void_p(n_reverts.data_ptr()) if n_reverts is not None else None))
        return mu, F, flag

    def gram(self, F, out=None, shift=0.0, shift_dev=None):
        """cov = F^T F (gsmvi_gram_f64): the covariance a square factor represents -- return value of the
        factor-form fit and what its monitor sees (gsm_numpy.py:129).  Not on the per-iteration path.
        ``shift`` / ``shift_dev`` (a 1-element float64 device tensor): cov = F^T F + (shift + shift_dev) I
        (gsmvi_gram_shift_f64) -- the jitter a factor-form BaM fit owes (bam.py:198)."""
        D = F.shape[0]
        assert F.shape == (D, D)
        self._ensure(D, max(self._max_B, 1))
        Cm = self.empty(D, D) if out is None else out
        pf, ldf = self._mat(F, "F")
        pc, ldc = self._mat(Cm, "C")
        if shift == 0.0 and shift_dev is None:
            _lib.check("gsmvi_gram_f64", self.lib.gsmvi_gram_f64(self._ctx, self._stream(), D, pf, ldf, pc, ldc))
        else:
            _lib.check("gsmvi_gram_shift_f64", self.lib.gsmvi_gram_shift_f64(
                self._ctx, self._stream(), D, pf, ldf, float(shift),
                C.c_void_p(shift_dev.data_ptr()) if shift_dev is not None else None, pc, ldc))
        return Cm

    def owed_shift(self, jitter, pend, n_rev, mark, advance=True):
        """The diagonal shift a factor-form BaM fit owes its covariance: jitter * (accepted updates since the last absorption)
        as a 1-element device tensor, without a host synchronisation.  ``pend`` = updates ATTEMPTED since then (host count),
        ``n_rev`` = the device counter of reverts, ``mark`` = its value at the last absorption (a device int32 word, advanced
        to ``n_rev`` when ``advance``): a reverted update adds no jitter in the reference (bam.py:198 sits before the accept
        test, :208-212 discards the shifted matrix with the update)."""
        s = (float(pend) - (n_rev - mark).to(torch.float64)) * float(jitter)
        if advance:
            mark.copy_(n_rev)
        return s

    def whiten_rows(self, X, mu, R):
        """(Z, logdiag): Z = (X - mu) R^-1 for the rows of X and logdiag = sum_i log R_ii (a 1-element device
        tensor); log N(x_b; mu, R^T R) = -|z_b|^2/2 - logdiag - D/2 log(2 pi)   [gsmvi/monitors.py:107]."""
        n, D = X.shape
        self._ensure(D, max(self._max_B, 1))
        Z = self.empty(n, D)
        ld = self.empty(1)
        px, ldx = self._mat(X, "X")
        pr, ldr = self._mat(R, "R")
        pz, ldz = self._mat(Z, "Z")
        _lib.check("gsmvi_whiten_rows_f64", self.lib.gsmvi_whiten_rows_f64(
            self._ctx, self._stream(), D, n, pr, ldr, px, ldx, self._vec(mu, "mu") if mu is not None else None,
            pz, ldz, C.c_void_p(ld.data_ptr())))
        return Z, ld

    # kernel-family bits of include/gsmvi_hip.h (GSMVI_PATH_*)
    PATH_BITS = {"panel_fast": 0x1, "panel_wide": 0x2, "panel_generic": 0x4, "panel_t_fast": 0x8, "panel_t_generic": 0x10,
                 "scalars_fast": 0x20, "scalars_generic": 0x40, "cov_sym": 0x80, "cov_generic": 0x100, "fupd_fast": 0x200,
                 "fupd_generic": 0x400, "lowrank_fast": 0x800, "lowrank_generic": 0x1000}
    PATH_GENERIC_MASK = 0x4 | 0x10 | 0x40 | 0x100 | 0x400 | 0x1000

    def last_path(self, reset=True):
        """Names of the kernel families launched on this context since the last reset (gsmvi_last_path): how tests and
        profiles check that an off-grid shape stayed on the tuned kernels (no name ending in ``_generic``)."""
        self._ensure(max(self._max_D, 1), max(self._max_B, 1))
        bits = C.c_uint(0)
        _lib.check("gsmvi_last_path", self.lib.gsmvi_last_path(self._ctx, C.byref(bits), int(bool(reset))))
        return {k for k, v in self.PATH_BITS.items() if bits.value & v}

    def bam_reg_source(self, word):
        """gsmvi_bam_set_reg_source: ``word`` (a 1-element float64 device tensor) makes every BaM update launched from now on
        read its regulariser from that word when its kernels EXECUTE (a captured graph can then be replayed with another
        value; the ``reg`` argument is ignored); ``None`` restores the by-value argument."""
        self._ensure(max(self._max_D, 1), max(self._max_B, 1))
        self._reg_word = word                                   # (kept alive while it is the source)
        ptr = None if word is None else C.c_void_p(word.data_ptr())
        _lib.check("gsmvi_bam_set_reg_source", self.lib.gsmvi_bam_set_reg_source(self._ctx, ptr))

    def set_profiling(self, on):
        self._ensure(max(self._max_D, 1), max(self._max_B, 1))
        _lib.check("gsmvi_set_profiling", self.lib.gsmvi_set_profiling(self._ctx, int(bool(on))))

    def get_profile(self):
        """Kernel durations (ms) of the last profiled update: panel, scalars, cov_update."""
        ms = (C.c_float * 3)()
        _lib.check("gsmvi_get_profile", self.lib.gsmvi_get_profile(self._ctx, ms, 3))
        return {"panel": ms[0], "scalars": ms[1], "cov_update": ms[2]}

    def gaussian_score(self, X, m, P, out=None):
        """G = -(X - m) P   [examples/example_gsm_numpy.py:24-29]."""
        B, D = X.shape
        self._ensure(D, B)
        G = self.empty(B, D) if out is None else out
        px, ldx = self._mat(X, "X")
        pp, ldp = self._mat(P, "P")
        pg, ldg = self._mat(G, "G")
        _lib.check("gsmvi_gaussian_score_f64", self.lib.gsmvi_gaussian_score_f64(
            self._ctx, self._stream(), D, B, px, ldx, self._vec(m, "m"), pp, ldp, pg, ldg))
        return G

    def potrf(self, S, out=None, flag=None):
        """Upper Cholesky factor R (R^T R = S) and a device flag (0 = positive definite)."""
        D = S.shape[0]
        self._ensure(D, max(self._max_B, 1))
        R = self.empty(D, D) if out is None else out
        flag = self.new_flag() if flag is None else flag
        ps, lds = self._mat(S, "S")
        pr, ldr = self._mat(R, "R")
        _lib.check("gsmvi_potrf_f64", self.lib.gsmvi_potrf_f64(
            self._ctx, self._stream(), D, ps, lds, pr, ldr, C.c_void_p(flag.data_ptr())))
        return R, flag

    def sample(self, Z, mu, R, out=None):
        """X = mu + Z R   [replaces np.random.multivariate_normal, gsmvi/gsm_numpy.py:116]."""
        B, D = Z.shape
        self._ensure(D, B)
        X = self.empty(B, D) if out is None else out
        pz, ldz = self._mat(Z, "Z")
        pr, ldr = self._mat(R, "R")
        px, ldx = self._mat(X, "X")
        _lib.check("gsmvi_sample_f64", self.lib.gsmvi_sample_f64(
            self._ctx, self._stream(), D, B, pz, ldz, self._vec(mu, "mu"), pr, ldr, px, ldx))
        return X

    def commit(self, flag, mu_new, S_new, mu, S, n_reverts=None):
        """In place: (mu, S) <- (mu_new, S_new) iff flag == 0   [gsmvi/gsm_numpy.py:121-125]."""
        D = mu.shape[0]
        self._ensure(D, max(self._max_B, 1))
        psn, ldsn = self._mat(S_new, "S_new")
        ps, lds = self._mat(S, "S")
        _lib.check("gsmvi_commit_f64", self.lib.gsmvi_commit_f64(
            self._ctx, self._stream(), D, C.c_void_p(flag.data_ptr()), self._vec(mu_new, "mu_new"), psn, ldsn,
            self._vec(mu, "mu"), ps, lds, C.c_void_p(n_reverts.data_ptr()) if n_reverts is not None else None))

    def bam_update(self, X, G, mu0, S0, reg, jitter=0.0, out=None, flag=None):
        """(mu, S) of BaM [gsmvi/bam.py:72-114]; S symmetrised, jitter on the diagonal."""
        assert X.dim() == 2 and G.dim() == 2            # bam.py:47-48
        B, D = X.shape
        if B > self.bam_max_batch:                      # a deterministic limit: never inside a retry loop
            raise ValueError(f"BaM update: batch size {B} exceeds the device chain's limit of {self.bam_max_batch}")
        self._ensure(D, B)
        mu, S = (self.empty(D), self.empty(D, D)) if out is None else out
        flag = self.new_flag() if flag is None else flag
        px, ldx = self._mat(X, "samples")
        pg, ldg = self._mat(G, "vs")
        ps0, lds0 = self._mat(S0, "S0")
        ps, lds = self._mat(S, "S")
        _lib.check("gsmvi_bam_update_f64", self.lib.gsmvi_bam_update_f64(
            self._ctx, self._stream(), D, B, px, ldx, pg, ldg, self._vec(mu0, "mu0"), ps0, lds0,
            float(reg), float(jitter), self._vec(mu, "mu"), ps, lds, C.c_void_p(flag.data_ptr())))
        return mu, S, flag


_ENGINES = {}


def get_engine(device=None):
    """Process-wide engine per device (created on first use; raises if the library or GPU is missing).
    ONE context = ONE workspace: calls through the shared engine must not run concurrently on two streams or two
    threads; give every extra stream / thread its own ``HipEngine(device)`` (as bench.py --in-flight does)."""
    _lib.load_library()
    _require_gpu()
    idx = torch.cuda.current_device() if device is None else (device if isinstance(device, int)
                                                              else (torch.device(device).index or 0))
    if idx not in _ENGINES:
        _ENGINES[idx] = HipEngine(idx)
    return _ENGINES[idx]
