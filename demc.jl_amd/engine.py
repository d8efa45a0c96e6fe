"""``HipEngine``: one ``demcz_handle`` (one GPU, one shard of chains) behind a small Python face.

The drivers in ``sampler.py`` talk to an engine through the methods below and nothing else;
the product only ever constructs ``HipEngine``.  (tests/ inject an oracle-backed engine with
the same methods to exercise the host logic, e.g. the world_size-2 gloo tests, on machines
without a GPU.)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DemczError
from .targets import is_device_target


def blocks_to_csr(blockindex, d):
    """DEMCopt.blockindex (1-based ranges / index vectors, DEMC.jl:29) is given here as a list
    of 0-based index sequences; returns CSR (offsets, indices) as int32 arrays."""
    blocks = [np.asarray(list(b), dtype=np.int64) for b in blockindex]
    for b in blocks:
        if b.size == 0 or b.min() < 0 or b.max() >= d:
            raise ValueError("block indices must be 0-based and within [0, d)")
    offs = np.zeros(len(blocks) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([b.size for b in blocks])
    idx = np.concatenate(blocks).astype(np.int32)
    return offs, idx


class _PinnedOwner:
    """Pinned host buffers detached from a handle: returned to the library's pool when nothing refers to them any more."""

    def __init__(self, L, bases):
        self._L, self._bases = L, [b for b in bases if b]

    def __del__(self):
        try:
            for b in self._bases:
                self._L.demcz_release_host_buffer(C.c_void_p(b))
        except Exception:
            pass
        self._bases = []


class HipEngine:
    """Device state of one shard: Z replica, current states, history, RNG position."""

    def __init__(self, *, N, d, K, Mcap, Gcap, blockindex, eps_scale, seed, target, chain_id0=0,
                 device_id=0, stream=None, lanes_per_chain=0):
        self._L = _lib.load()
        self.N, self.d, self.K, self.Mcap, self.Gcap = int(N), int(d), int(K), int(Mcap), int(Gcap)
        self.chain_id0 = int(chain_id0)
        self._keep = []
        offs, idx = blocks_to_csr(blockindex, d)
        eps = _lib.f64(eps_scale)
        if eps.shape != (d,):
            raise ValueError("eps_scale must have d entries")
        self._keep += [offs, idx, eps]
        cfg = _lib.Config()
        cfg.N, cfg.chain_id0, cfg.d, cfg.K = self.N, self.chain_id0, self.d, self.K
        cfg.Mcap, cfg.Gcap, cfg.Nblocks = self.Mcap, self.Gcap, len(offs) - 1
        cfg.block_offsets, cfg.block_indices = _lib.ptr(offs, _lib._ip), _lib.ptr(idx, _lib._ip)
        cfg.eps_scale, cfg.seed, cfg.device_id = _lib.ptr(eps), int(seed) & (2**64 - 1), int(device_id)
        cfg.stream = stream
        cfg.lanes_per_chain = int(lanes_per_chain)
        if is_device_target(target):
            if target.d != d:
                raise ValueError("target dimension != d")
            cfg.target_kind = target.kind
            target.fill(cfg, self._keep)
        elif callable(target):
            cfg.target_kind = _lib.TARGET_HOST_CALLBACK
        else:
            raise TypeError("target must be a device target or a callable")
        self.host_callback = cfg.target_kind == _lib.TARGET_HOST_CALLBACK
        self._h = C.c_void_p()
        rc = self._L.demcz_create(C.byref(self._h), C.byref(cfg))
        if rc != 0:
            raise DemczError(rc, (self._L.demcz_last_error(None) or b"").decode())

    # -- plumbing -----------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise DemczError(rc, (self._L.demcz_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.demcz_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state --------------------------------------------------------------------------------
    def set_state(self, X, logp, Z):
        X = _lib.f64(X, "F")
        Z = _lib.f64(Z, "F")
        if X.shape != (self.N, self.d) or Z.ndim != 2 or Z.shape[1] != self.d:
            raise ValueError("X must be N x d and Z must be M0 x d")
        lp = _lib.f64(logp) if logp is not None else None
        if lp is not None and lp.shape != (self.N,):
            raise ValueError("logp must have N entries")
        self._chk(self._L.demcz_set_state(self._h, _lib.ptr(X), _lib.ptr(lp), _lib.ptr(Z), Z.shape[0], Z.shape[0]))

    def get_state(self, with_Z=True):
        X = np.empty((self.N, self.d), order="F")
        lp = np.empty(self.N)
        M = C.c_int64()
        self._chk(self._L.demcz_get_state(self._h, _lib.ptr(X), _lib.ptr(lp), None, 0, C.byref(M)))
        Z = None
        if with_Z:
            # the archive comes back in a pinned buffer of the library's pool (no page faults of a fresh 41 MB array under the
            # copy); the array below is a view of it, and the buffer returns to the pool with the array
            pz, Mz = C.c_void_p(), C.c_int64()
            self._chk(self._L.demcz_get_archive_pinned(self._h, C.byref(pz), C.byref(Mz)))
            buf = (C.c_double * (Mz.value * self.d)).from_address(pz.value)
            buf._owner = _PinnedOwner(self._L, [pz.value])
            Z = np.ctypeslib.as_array(buf).reshape((Mz.value, self.d), order="F")
        return X, lp, Z, int(M.value)

    @property
    def M(self):
        M = C.c_int64()
        self._chk(self._L.demcz_get_info(self._h, C.byref(M), None, None))
        return int(M.value)

    def info(self):
        M, nl, lanes = C.c_int64(), C.c_int64(), C.c_int32()
        self._chk(self._L.demcz_get_info(self._h, C.byref(M), C.byref(nl), C.byref(lanes)))
        return dict(M=int(M.value), window_launches=int(nl.value), lanes_per_chain=int(lanes.value))

    def set_append_lag(self, batches):
        self._chk(self._L.demcz_set_append_lag(self._h, int(batches)))

    def set_rng_offset(self, generations):
        self._chk(self._L.demcz_set_rng_offset(self._h, int(generations)))

    def set_history_origin(self, g0):
        self._chk(self._L.demcz_set_history_origin(self._h, int(g0)))

    # -- the hot path -------------------------------------------------------------------------
    def run(self, g_from, g_to, gamma, temperature=None):
        t = None
        if temperature is not None:
            t = _lib.f64(temperature)
            if t.shape != (g_to - g_from + 1,):
                raise ValueError("one temperature per generation")
        self._chk(self._L.demcz_run(self._h, int(g_from), int(g_to), float(gamma), _lib.ptr(t)))

    def run_checked(self, g_from, g_to, gamma, every, threshold=0.0, temperature=None):
        """demcz_run_checked: the generation loop with its R-hat test every ``every`` generations as one
        library call (demcz.jl:30-55).  Returns (g_stop, max R-hat of every check made, R-hat vector of the last)."""
        t = None
        if temperature is not None:
            t = _lib.f64(temperature)
            if t.shape != (g_to - g_from + 1,):
                raise ValueError("one temperature per generation")
        n_max = int((g_to - g_from + 1) // every + 2)
        mx = np.zeros(n_max)
        last = np.full(self.d, np.nan)
        g_stop, n = C.c_int64(0), C.c_int32(0)
        self._chk(self._L.demcz_run_checked(self._h, C.c_int64(int(g_from)), C.c_int64(int(g_to)), C.c_double(float(gamma)),
                                            _lib.ptr(t), C.c_int64(int(every)), C.c_double(float(threshold)), C.byref(g_stop),
                                            C.byref(n), _lib.ptr(mx), C.c_int32(n_max), _lib.ptr(last)))
        return int(g_stop.value), mx[:n.value].copy(), last

    def set_kernel_timing(self, enabled: bool):
        self._chk(self._L.demcz_set_kernel_timing(self._h, C.c_int32(1 if enabled else 0)))

    def get_kernel_time(self):
        """(launches, milliseconds) of the window kernels timed since the last call (HIP events on the kernel's stream)."""
        n, ms = C.c_int64(0), C.c_double(0.0)
        self._chk(self._L.demcz_get_kernel_time(self._h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def get_kernel_time_series(self):
        """(start_ms, duration_ms) arrays of the brackets the last get_kernel_time() summed up: one per demcz_run call."""
        n = C.c_int32(0)
        self._chk(self._L.demcz_get_kernel_time_series(self._h, 0, None, None, C.byref(n)))
        st, du = np.zeros(n.value), np.zeros(n.value)
        self._chk(self._L.demcz_get_kernel_time_series(self._h, n.value, _lib.ptr(st), _lib.ptr(du), C.byref(n)))
        return st, du

    def set_live_spin_limit(self, polls: int):
        """Diagnostic: polls before a LIVE row wait gives up (tests force the fall-back path with 1)."""
        self._chk(self._L.demcz_set_live_spin_limit(self._h, C.c_int32(int(polls))))

    def debug_set_live_fault(self, polls: int, g_from: int = 0):
        """Diagnostic: LIVE launches starting at generation g_from or later use the poll limit `polls` (0: off)."""
        self._chk(self._L.demcz_debug_set_live_fault(self._h, C.c_int32(int(polls)), C.c_int64(int(g_from))))

    def live_status(self):
        """Diagnostic: (LIVE launches in use, number of fall-backs to one launch per K-window)."""
        on, redos = C.c_int32(0), C.c_int32(0)
        self._chk(self._L.demcz_get_live_status(self._h, C.byref(on), C.byref(redos)))
        return bool(on.value), int(redos.value)

    def set_live_rearms(self, n: int):
        """How many more times the handle may go back to LIVE launches after a failed hand-off (default 3; 0: a failure leaves
        it at one launch per K-window for good, the behaviour of rounds 2-4)."""
        self._chk(self._L.demcz_set_live_rearms(self._h, C.c_int32(int(n))))

    def live_rearms(self):
        """Diagnostic: (times the handle went LIVE again after a fall-back, re-arms left)."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._chk(self._L.demcz_get_live_rearms(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def kernel_counts(self):
        """Diagnostic: window launches so far by kernel -- {"ps2", "ps_general", "other"} (demcz_debug_kernel_counts)."""
        c = (C.c_int64 * 3)()
        self._chk(self._L.demcz_debug_kernel_counts(self._h, c))
        return dict(zip(("ps2", "ps_general", "other"), (int(v) for v in c)))

    def kernel_name(self):
        """Diagnostic: the window kernel the most recent window launch ran, as a profiler names it."""
        buf = C.create_string_buffer(192)
        self._chk(self._L.demcz_debug_kernel_name(self._h, buf, 192))
        return buf.value.decode()

    def synchronize(self):
        self._chk(self._L.demcz_synchronize(self._h))

    # -- results ------------------------------------------------------------------------------
    def get_history(self, g_from, g_to, chain=True, log_obj=True, out=None):
        """`out`: (chain, log_obj) column-major arrays of at least g_to - g_from + 1 generations to fill instead of new ones
        (their pages already touched: the download then runs at the link's rate, not at the page faults'); the views of
        the generations asked for are returned."""
        G = g_to - g_from + 1
        if out is not None and chain and log_obj:
            ch_full, lo_full = out
            ok = (ch_full.flags.f_contiguous and lo_full.flags.f_contiguous and ch_full.shape[:2] == (self.N, self.d)
                  and lo_full.shape[0] == self.N and ch_full.shape[2] >= G and lo_full.shape[1] >= G
                  and ch_full.dtype == np.float64 and lo_full.dtype == np.float64)
            if ok:
                self._chk(self._L.demcz_get_history(self._h, int(g_from), int(g_to), _lib.ptr(ch_full), _lib.ptr(lo_full)))
                return ch_full[:, :, :G], lo_full[:, :G]
        ch = np.empty((self.N, self.d, G), order="F") if chain else None
        lo = np.empty((self.N, G), order="F") if log_obj else None
        self._chk(self._L.demcz_get_history(self._h, int(g_from), int(g_to), _lib.ptr(ch), _lib.ptr(lo)))
        return ch, lo

    # -- streamed history: pinned host mirrors filled while the GPU runs (demcz_history_stream) -------------------------
    def history_stream(self, enabled=True):
        self._chk(self._L.demcz_history_stream(self._h, 1 if enabled else 0))
        self._streaming = bool(enabled)

    def take_history(self, g_from, g_to):
        """mc.chain[:, :, g_from:g_to], mc.log_obj[:, g_from:g_to] as NumPy arrays OVER the library's pinned mirrors (no copy):
        the mirrors are detached from the handle and go back to the library's pool when the last array over them is
        collected.  One call per run (the handle keeps no mirrors afterwards)."""
        G = g_to - g_from + 1
        pc, pl = C.c_void_p(), C.c_void_p()
        self._chk(self._L.demcz_get_history_view(self._h, int(g_from), int(g_to), C.byref(pc), C.byref(pl)))
        bc, bl = C.c_void_p(), C.c_void_p()
        self._chk(self._L.demcz_detach_history(self._h, C.byref(bc), C.byref(bl)))
        self._streaming = False
        owner = _PinnedOwner(self._L, [bc.value, bl.value])
        cbuf = (C.c_double * (self.N * self.d * G)).from_address(pc.value)
        lbuf = (C.c_double * (self.N * G)).from_address(pl.value)
        cbuf._owner = owner          # (NumPy keeps the ctypes object alive; the ctypes object keeps the owner alive)
        lbuf._owner = owner
        chain = np.ctypeslib.as_array(cbuf).reshape((self.N, self.d, G), order="F")
        lobj = np.ctypeslib.as_array(lbuf).reshape((self.N, G), order="F")
        return chain, lobj

    def get_changed(self, g_from, g_to):
        out = np.zeros(g_to - g_from + 1, dtype=np.int64)
        self._chk(self._L.demcz_get_changed(self._h, int(g_from), int(g_to), _lib.ptr(out, _lib._lp)))
        return out

    def changed_total(self, g_from, g_to, with_source=False):
        """Sum of get_changed over g_from..g_to, from the window kernels' ballot counters when the range is made of
        whole launches (demcz_get_changed_total).  with_source: also return whether the ballots answered."""
        tot, src = C.c_int64(0), C.c_int32(0)
        self._chk(self._L.demcz_get_changed_total(self._h, int(g_from), int(g_to), C.byref(tot), C.byref(src)))
        return (int(tot.value), bool(src.value)) if with_source else int(tot.value)

    def rhat(self, g_from, g_to):
        out = np.empty(self.d)
        self._chk(self._L.demcz_rhat(self._h, int(g_from), int(g_to), _lib.ptr(out)))
        return out

    def rhat_partial(self, g_from, g_to, stage, grand):
        out = np.empty(self.d if stage == 0 else 2 * self.d)
        g = _lib.f64(grand) if grand is not None else None
        self._chk(self._L.demcz_rhat_partial(self._h, int(g_from), int(g_to), int(stage), _lib.ptr(g), _lib.ptr(out)))
        return out

    def accept_ratio(self, g_from, g_to):
        out = np.empty(self.N)
        self._chk(self._L.demcz_accept_ratio(self._h, int(g_from), int(g_to), _lib.ptr(out)))
        return out

    def mean_cov(self, g_from, g_to):
        mean = np.empty(self.d)
        cov = np.empty((self.d, self.d), order="F")
        self._chk(self._L.demcz_mean_cov(self._h, int(g_from), int(g_to), _lib.ptr(mean), _lib.ptr(cov)))
        return mean, cov

    # -- host-closure mode ----------------------------------------------------------------------
    def closure_buffers(self):
        """demcz_closure_buffers: (Xprop, logp) NumPy views of the handle's pinned host buffers -- N x d column-major and N.
        After this call propose() returns a view of Xprop (no copy, no stream synchronisation) and accept_commit() with no
        argument commits whatever has been written into logp."""
        if getattr(self, "_hc", None) is None:
            px, pl = C.c_void_p(), C.c_void_p()
            self._chk(self._L.demcz_closure_buffers(self._h, C.byref(px), C.byref(pl)))
            bx = (C.c_double * (self.N * self.d)).from_address(px.value)
            bl = (C.c_double * self.N).from_address(pl.value)
            self._hc = (np.ctypeslib.as_array(bx).reshape((self.N, self.d), order="F"), np.ctypeslib.as_array(bl))
        return self._hc

    def propose(self, g, ib, gamma):
        if getattr(self, "_hc", None) is not None:
            self._chk(self._L.demcz_propose(self._h, int(g), int(ib), float(gamma), None))
            return self._hc[0]
        Xp = np.empty((self.N, self.d), order="F")
        self._chk(self._L.demcz_propose(self._h, int(g), int(ib), float(gamma), _lib.ptr(Xp)))
        return Xp

    def accept_commit(self, logp_prop=None, temperature=None):
        t = C.byref(C.c_double(float(temperature))) if temperature is not None else None
        tp = C.cast(t, _lib._dp) if t is not None else None
        hc = getattr(self, "_hc", None)
        if logp_prop is None or (hc is not None and logp_prop is hc[1]):
            if hc is None:
                raise ValueError("accept_commit() without values needs closure_buffers() first")
            self._chk(self._L.demcz_accept_commit(self._h, None, tp))
            return
        lp = _lib.f64(logp_prop)
        self._chk(self._L.demcz_accept_commit(self._h, _lib.ptr(lp), tp))

    def end_generation(self, g):
        self._chk(self._L.demcz_end_generation(self._h, int(g)))

    # -- multi-GPU ------------------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self._L.demcz_comm_unique_id(buf)
        if rc != 0:
            raise DemczError(rc, (self._L.demcz_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        buf = C.create_string_buffer(unique_id, 128)
        self._chk(self._L.demcz_comm_init(self._h, buf, int(nranks), int(rank)))

    @staticmethod
    def peer_group(engines):
        """demcz_peer_group: the engines (shards of one run, all in this process on one device) publish their K-boundary rows
        into each other's archive replicas from inside their launches."""
        L = engines[0]._L
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        rc = L.demcz_peer_group(arr, len(engines))
        if rc != 0:
            msgs = [(L.demcz_last_error(e._h) or b"").decode() for e in engines]
            raise DemczError(rc, next((m for m in msgs if m), ""))

    def peer_export(self, nranks: int, rank: int) -> bytes:
        """demcz_peer_export: the archive becomes fine-grained, IPC-exportable memory; returns its 64-byte handle."""
        buf = C.create_string_buffer(64)
        self._chk(self._L.demcz_peer_export(self._h, int(nranks), int(rank), buf))
        return buf.raw

    def peer_attach(self, handles):
        """demcz_peer_attach: `handles` = the 64-byte handles of all ranks in rank order (this rank's own included)."""
        blob = b"".join(handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._chk(self._L.demcz_peer_attach(self._h, buf))

    def peer_detach(self):
        """demcz_peer_detach (host-mediated IPC peers): close this rank's mappings of the other ranks' archives.  Call between
        two barriers of the host's, after the last run and before any rank closes its engine."""
        self._chk(self._L.demcz_peer_detach(self._h))

    def peer_ping(self):
        """(ok, wait_us) of demcz_comm_init's first-contact check: ok = 1 passed on all ranks, 0 failed somewhere (the run
        exchanges through ncclAllGather), -1 not made; wait_us = how long this rank waited for the last peer's token."""
        ok, us = C.c_int32(-1), C.c_double(0.0)
        self._chk(self._L.demcz_get_peer_ping(self._h, C.byref(ok), C.byref(us)))
        return int(ok.value), float(us.value)

    def peer_status(self):
        """(mode, peers): mode 0 = RCCL exchange or unsharded, 1 = replica group of this process (demcz_peer_group), 2 = IPC
        peers set up over the library's communicator (demcz_comm_init), 3 = IPC peers set up by the host carrying the handles
        (demcz_peer_export / demcz_peer_attach: no automatic redo, the host owes the barriers); peers = replicas this handle
        publishes into besides its own."""
        m, n = C.c_int32(0), C.c_int32(0)
        self._chk(self._L.demcz_get_peer_status(self._h, C.byref(m), C.byref(n)))
        return int(m.value), int(n.value)

    def set_comm_timeout(self, milliseconds: int):
        """Deadline of every host-side wait of a sharded handle; past it the communicators are aborted and calls raise
        DemczError with code ERR_COMM (0 = wait for ever)."""
        self._chk(self._L.demcz_set_comm_timeout(self._h, C.c_int64(int(milliseconds))))

    def debug_stall_exchange(self, milliseconds: int):
        """Diagnostic: hold the next collective back on its stream for `milliseconds` (what a stalled peer looks like)."""
        self._chk(self._L.demcz_debug_stall_exchange(self._h, C.c_int32(int(milliseconds))))

    def set_external_append(self, enabled: bool):
        self._chk(self._L.demcz_set_external_append(self._h, 1 if enabled else 0))

    def append_rows(self, rows):
        rows = _lib.f64(rows, "F")
        if rows.ndim != 2 or rows.shape[1] != self.d:
            raise ValueError("rows must be nrows x d")
        self._chk(self._L.demcz_append_rows(self._h, _lib.ptr(rows), rows.shape[0], rows.shape[0]))

    def debug_append_slab(self, slab, R, cnt, batched):
        """Diagnostic: run the sharded exchange's scatter kernel on a caller-built [R][cnt][d][N] slab."""
        slab = _lib.f64(slab)
        if slab.size != R * cnt * self.d * self.N:
            raise ValueError("slab must hold R*cnt*d*N doubles")
        self._chk(self._L.demcz_debug_append_slab(self._h, _lib.ptr(slab), int(R), int(cnt), 1 if batched else 0))

    def export_current_device(self, device_ptr: int):
        self._chk(self._L.demcz_export_current_device(self._h, C.c_void_p(device_ptr)))

    def append_rows_device(self, device_ptr: int, nrows: int, ldrows: int):
        self._chk(self._L.demcz_append_rows_device(self._h, C.c_void_p(device_ptr), int(nrows), int(ldrows)))


def pool_trim():
    """Give the buffers the library keeps for the next handle back to the runtime: (device bytes, pinned bytes) freed."""
    L = _lib.load()
    a, b = C.c_int64(0), C.c_int64(0)
    L.demcz_pool_trim(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def selftest_draws(seed, chain, blk0, n, device_id=0):
    """Device draw pipeline for blocks blk0..blk0+n-1 of chain's stream (see demcz.h)."""
    L = _lib.load()
    words = np.zeros(2 * n, dtype=np.uint64)
    normals = np.zeros(2 * n)
    logu = np.zeros(n)
    rc = L.demcz_selftest_draws(int(device_id), int(seed), int(chain), int(blk0), int(n),
                                words.ctypes.data_as(C.POINTER(C.c_uint64)), _lib.ptr(normals), _lib.ptr(logu))
    if rc != 0:
        raise DemczError(rc, (L.demcz_last_error(None) or b"").decode())
    return words.reshape(n, 2), normals.reshape(n, 2), logu
