"""Sharded (multi-GPU) host path: one process per GPU, torch.distributed over RCCL.

Partitioning: every rank owns a contiguous slice of the record stream and
runs K1 on it; a key belongs to shard hash(cell_index) mod G, so all UMIs of a
(cell, feature) group meet on one GPU.  Steps of one pass:

  1. K0 hit count of the local slice → all_gather (one u64 per rank) → draw-rank base,
     so that the keep/drop decision of record i uses the same MT draw as in a serial run
     (bam2db_ds.c:385 is consumed only by CB hits, in record order);
  2. K1 probe/filter/pack, keys written straight into one buffer per destination shard;
  3. ONE exchange of the keys (xGMI).  The first step of a ShardedPass learns the per-destination key counts the exact
     way (all_to_all of the counts, a host synchronisation for the split sizes, all_to_all_single with splits); every
     later step runs the FIXED-CAPACITY form: each shard buffer is a row of `cap` key slots + one header slot that carries
     its count, the exchange is one all_to_all_single with equal splits, and the receiver sorts straight out of the G
     received rows (the engine's region map, as for the streaming K1b) — two collectives per step (the hit counts, the
     keys), no host synchronisation, nothing on the host that depends on device results.  A count beyond `cap` shows in
     the headers; ensure_exact() (always run before results are read) then repeats the step the exact way;
  4. local K2 sort + K3 reduce on the received keys;
  5. COO rows stay on their shard (gather_coo() merges them); the three counters are summed over ranks on demand
     (global_counters(): one all_reduce, outside the data path).

Keys wider than 64 bits (a raw whitelist x 36 k genes x 12-base UMIs needs 66; UMIs beyond 16 bases always): stages with
``wide`` set run the same five steps with the key in two words (``_run_wide``): K1 writes the group word and the rest of the key
into two buffers per destination, the exact protocol exchanges both (the counts, then two all_to_all_single with splits), the
receiver hands the pairs to its engine (``reduce_wide``: sort on the group word with the values riding along, reduce, rows).  No
fixed-capacity form and no pipelining there: the path exists so that no input the single-GPU engine takes is refused by the
sharded one; its throughput is not the headline's.

With more than one shard on GPUs the pass is software-pipelined over three HIP streams: the K1 stage of step i+1
(hit count, draw-rank base, probe/filter/pack, count exchange) on one, the key exchange of step i on another, sort and
reduce on the caller's stream — K1 and the sort are HBM-bound, the exchange is xGMI-bound, so the exchange overlaps
both; the buffers K1 writes (shard buffers, counters) and the receive buffer are double-buffered, and events keep K1
from overwriting a shard buffer that is still being sent and the exchange from landing in a receive buffer that is still
being sorted.  In that mode the two tiny collectives of the K1 stage travel as host integers over
a gloo side group (RCCL would queue them behind the running key exchange, and the host needs the counts anyway).
``FASTF_DIST_PIPELINE=0`` runs everything on one stream with every collective on the default group.

The device stages are injected (``stages``): the default is the HIP engine; the CPU test
suite drives the same orchestration over gloo with test doubles.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from .hostmem import copy_into, to_device, to_host, to_host_tensor


def owner_of_cell(cell_index: np.ndarray, n_shards: int) -> np.ndarray:
    """host mirror of shard_of() in umi_kernels.hpp (murmur3 finaliser, top 32 bits, mod G)"""
    x = cell_index.astype(np.uint64)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xc4ceb9fe1a85ec53)
    x ^= x >> np.uint64(33)
    return ((x >> np.uint64(32)) % np.uint64(n_shards)).astype(np.int64)


RUN_TOO_LONG = 16      # FASTF_ERR_RUN_TOO_LONG


class DrawBits:
    """the decisions of a draw stream on the device (HipStages.draw_bits): bit i = draw i < the engine's threshold"""

    def __init__(self, words, n_draws):
        self.words, self.n_draws = words, n_draws


class HipStages:
    """device stages backed by libfastf_amd.so (fastf_dev_* entry points)"""

    def __init__(self, engine, device):
        self.eng, self.device = engine, device
        self.skip_low = True       # matrix-only sort: low digit passes skipped (switched off if the data has long runs)

    def run_too_long(self):
        """True if the last group-only reduce met an unsorted run beyond its cap; clears the flag (synchronises)"""
        if not self.skip_low:
            return False
        if self.eng.dev_error_bits() & RUN_TOO_LONG:
            self.eng.dev_clear_error_bits(RUN_TOO_LONG, self._s())
            return True
        return False

    def _s(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    @property
    def wide(self):
        return self.eng.wide

    def probe_pack_wide(self, cb, gx, umi, meta, ext, n, draws, draw_base, keys_out, vals_out, stride, key_counts, counters, reuse_hits=False):
        """K1 of an engine whose keys are wider than 64 bits: group words into keys_out[shard], the rest of each key into
        vals_out[shard] (fastf_dev_probe_pack_wide); ext: bases 17.. of the UMIs or None"""
        bits = isinstance(draws, DrawBits)
        self.eng.dev_probe_pack_wide(cb.data_ptr(), gx.data_ptr(), umi.data_ptr(), meta.data_ptr(), 0 if ext is None else ext.data_ptr(), n,
                                     draws.words.data_ptr() if bits else draws.data_ptr(), draws.n_draws if bits else draws.numel(),
                                     keys_out.data_ptr(), vals_out.data_ptr(), stride, key_counts.data_ptr(), counters.data_ptr(), self._s(),
                                     d_draw_base=draw_base.data_ptr(), reuse_hits=reuse_hits, draw_bits=bits)

    def reduce_wide(self, keys, vals, n):
        """the n pairs this shard owns -> its rows (feature, cell, count) on the host, ascending (cell, feature):
        fastf_dev_adopt_wide + fastf_engine_finish (K2 on the group word, K3 on the values)"""
        self.eng.dev_adopt_wide(keys.data_ptr(), vals.data_ptr(), n, self._s())
        r = self.eng.finish()
        return r["feature"].astype(np.int64), r["cell"].astype(np.int64), r["count"].astype(np.int64)

    def block(self, gx, umi, meta, n):
        """the records' gx / umi / meta in the engine's BLOCKED layout (one contiguous run per 256-record unit, with the
        slice K1a fills): pass the result as `gx` (umi = meta = None) to count_hits / probe_pack.  None when this engine has
        no streaming K1b to read it."""
        nb = self.eng.block_bytes(n)
        if not nb:
            return None
        buf = torch.empty(nb, dtype=torch.uint8, device=self.device)
        self.eng.dev_block_records(gx.data_ptr(), umi.data_ptr(), meta.data_ptr(), n, buf.data_ptr(), self._s())
        return buf

    def count_hits(self, cb, n, out_hits, blocked=None):
        if blocked is not None:
            self.eng.dev_count_hits_blocked(cb.data_ptr(), n, blocked.data_ptr(), out_hits.data_ptr(), self._s())
        else:
            self.eng.dev_count_hits(cb.data_ptr(), n, out_hits.data_ptr(), self._s())

    def key_slots(self, n, n_shards):
        """key slots per shard buffer: the streaming K1b of a single-shard pass writes one region per workgroup"""
        if n_shards == 1:
            return max(n, self.eng.probe_capacity(n))
        return n

    def draw_bits(self, draws):
        """K1b reads one bit per CB hit (kept or not); a stream of 32-bit draws that is used for more than one step is turned
        into those bits ONCE (fastf_dev_draw_bits) and the result passed to probe_pack in its place"""
        n = draws.numel()
        words = torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=self.device)
        self.eng.dev_draw_bits(draws.data_ptr(), n, words.data_ptr(), self._s())
        return DrawBits(words, n)

    def mt_decisions(self, seed, skip, n, out=None):
        """DrawBits of n draws of init_genrand(seed) + skip made by the device's generator (fastf_dev_mt_decisions); `out`: the
        DrawBits of an earlier call of at least that size, written again"""
        words = out.words if out is not None else torch.zeros((n + 63) // 64 * 2 + 2, dtype=torch.int32, device=self.device)
        self.eng.dev_mt_decisions(seed, skip, n, words.data_ptr(), self._s())
        return DrawBits(words, n)

    def probe_pack(self, cb, gx, umi, meta, n, draws, draw_base, keys_out, stride, key_counts, counters, reuse_hits=False):
        """draws: a tensor of 32-bit draws (converted on every call) or the DrawBits made from one"""
        self.segmented = self.eng.n_shards == 1 and self.eng.probe_capacity(n) > 0 and stride >= self.eng.probe_capacity(n)
        blocked = umi is None                              # gx is then a buffer made by block()
        bits = isinstance(draws, DrawBits)
        self.eng.dev_probe_pack(cb.data_ptr(), gx.data_ptr(), 0 if blocked else umi.data_ptr(), 0 if blocked else meta.data_ptr(), n,
                                draws.words.data_ptr() if bits else draws.data_ptr(), draws.n_draws if bits else draws.numel(),
                                keys_out.data_ptr(), stride,
                                key_counts.data_ptr(), counters.data_ptr(), self._s(),
                                d_draw_base=draw_base.data_ptr(), reuse_hits=reuse_hits, segmented=self.segmented, blocked=blocked,
                                draw_bits=bits)

    def set_regions(self, counts, n_regions, stride, d_n):
        """the keys of the next sort_reduce lie in n_regions rows of `stride` slots, row r holding counts[r] keys (device
        tensor); writes the total to d_n on the device"""
        self.eng.dev_set_regions(counts.data_ptr(), n_regions, stride, d_n.data_ptr(), self._s())
        self.segmented = True

    def sort_reduce(self, keys, tmp, d_n, max_n, feature, cell, count, nnz, fresh=True):
        # matrix only: the low digit passes are skipped, K3 resolves the short unsorted runs.
        # fresh: keys come straight from probe_pack (possibly segmented); a re-sort of already sorted keys is contiguous
        in_tmp = self.eng.dev_sort(keys.data_ptr(), tmp.data_ptr(), d_n.data_ptr(), max_n, stream=self._s(),
                                   skip_low=self.skip_low, segmented=fresh and getattr(self, "segmented", False))
        src = tmp if in_tmp else keys
        # K3 leaves the rows in the engine's row regions (one per workgroup chunk) and reports the count; rows_gather
        # concatenates them into feature/cell/count when somebody wants them (ShardedPass.local_coo)
        self.eng.dev_reduce(src.data_ptr(), d_n.data_ptr(), max_n, feature.data_ptr(), cell.data_ptr(),
                            count.data_ptr(), nnz.data_ptr(), self._s(), skip_low=self.skip_low, segmented=True)
        return src

    def rows_gather(self, d_n, feature, cell, count):
        self.eng.dev_rows_gather(d_n.data_ptr(), feature.data_ptr(), cell.data_ptr(), count.data_ptr(), self._s())


class ShardedPass:
    """Buffers + orchestration of one sharded pass; reusable across steps (bench loop)."""

    def __init__(self, stages, n_local_max, device, group=None, world=None, rank=None, pipeline=None):
        self.st, self.dev, self.group = stages, device, group
        self.G = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        # gloo has no all_to_all on device tensors: with that backend and HBM buffers every collective is staged
        # through host memory (rehearsal of the N>1 path on a box whose ranks share one GPU; RCCL refuses that)
        self.host_staged = (dist.is_initialized() and dist.get_backend(group) != "nccl"
                            and torch.device(device).type == "cuda")
        self._nccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
        G, n = self.G, int(n_local_max)
        i64, i32 = torch.int64, torch.int32
        self.stride = stages.key_slots(n, G) if hasattr(stages, "key_slots") else n
        if pipeline is None:
            pipeline = os.environ.get("FASTF_DIST_PIPELINE", "1") != "0"
        self.pipelined = bool(pipeline) and G > 1 and torch.device(device).type == "cuda"
        self.k1_stream = torch.cuda.Stream(device) if self.pipelined else None
        self.x_stream = torch.cuda.Stream(device) if self.pipelined else None     # the key exchange
        # The two tiny collectives of the K1 stage (one u64 per rank, G counts per rank):
        #   RCCL    device tensors on the SAME communicator as the key exchange, issued from the K1 stream.  One
        #           communicator runs its collectives in issue order, so the K1 stage of step i+1 starts behind the key
        #           exchange of step i and overlaps that step's sort; the host synchronises once per step, to learn the
        #           exchange sizes.  (No side group: a second communicator next to a running exchange is how RCCL jobs
        #           deadlock, and a gloo group costs two host round trips per step.)
        #   gloo    (CPU test suite, one-GPU rehearsal) host integers on the group itself.
        self.small_group = group if (self.pipelined and dist.is_initialized() and dist.get_backend(group) == "gloo") else None
        self.host_small = self.small_group is not None
        nb = 2 if self.pipelined else 1
        # what K1 writes per step, one set per pipeline slot.  The per-step scalars live in one buffer so a step clears
        # them with a single fill, but 512 B apart: atomics (key_counts, counters) and the plain loads of draw_base
        # must not share a cache line
        self._keys_out_slots = [torch.empty((G, self.stride), dtype=i64, device=device) for _ in range(nb)]
        self._small_slots = [torch.zeros(192, dtype=i64, device=device) for _ in range(nb)]
        self._slot_free = [None] * nb          # event: the exchange that read this slot's shard buffers has finished
        self._step = 0
        self._use_slot(0)
        self.recv_counts = torch.zeros(G, dtype=i64, device=device)
        self.hits = torch.zeros(1, dtype=i64, device=device)
        self.all_hits = torch.zeros(G, dtype=i64, device=device)
        self.recv_cap = G * n
        # the exchange of step i+1 may land while step i is still being sorted: two receive buffers when pipelined
        self._recv_slots = [torch.empty(self.recv_cap, dtype=i64, device=device) for _ in range(nb)]
        self._recv_free = [None] * nb          # event: sort + reduce of the step that used this receive buffer are done
        self.recv = self._recv_slots[0]
        self.tmp = torch.empty(self.recv_cap, dtype=i64, device=device)
        self._d_n_buf = torch.zeros(1, dtype=i64, device=device)
        self.d_n = self._d_n_buf
        self.feature = torch.empty(self.recv_cap, dtype=i32, device=device)
        self.cell = torch.empty(self.recv_cap, dtype=i32, device=device)
        self.count = torch.empty(self.recv_cap, dtype=i32, device=device)
        self.nnz = torch.zeros(1, dtype=i64, device=device)
        self.n_recv = 0
        self.sorted = None
        self._verified = False     # the group-only sort was checked against its run cap for the current result
        self._counters_reduced = True
        # fixed-capacity exchange (see the module docstring): key slots per (source, destination) pair, learned from the
        # first step; None = the next step runs the exact protocol
        self.cap = None
        self.fixed = os.environ.get("FASTF_DIST_FIXED", "1") != "0" and hasattr(stages, "set_regions")
        self._fixed_step = False   # the last step ran the fixed-capacity form
        self._overflow = None      # device flag of that step: some count exceeded cap
        self._last_inputs = None
        self.n_collectives = 0     # collectives issued so far (tests assert two per fixed-capacity step)

    def _use_slot(self, b):
        G = self.G
        self.keys_out = self._keys_out_slots[b]
        self._small = self._small_slots[b]
        self.key_counts = self._small[:G]
        self.counters = self._small[64:68]
        self.draw_base = self._small[128:129]

    def _k1_fixed(self, cb, gx, umi, meta, n, draws):
        """fixed-capacity step: hit count → all_gather → draw-rank base → probe/filter/pack into rows of cap + 1 slots →
        the counts into the rows' header slots.  Nothing comes back to the host."""
        G, st, cap = self.G, self.st, self.cap
        self._small.zero_()
        if umi is None:
            st.count_hits(cb, n, self.hits, blocked=gx)
        else:
            st.count_hits(cb, n, self.hits)
        self._all_gather(self.all_hits, self.hits)
        self.draw_base.copy_(self.all_hits[:self.rank].sum().reshape(1))
        rows = self.keys_out.view(-1)[:G * (cap + 1)].view(G, cap + 1)
        st.probe_pack(cb, gx, umi, meta, n, draws, self.draw_base, rows, cap + 1, self.key_counts, self.counters, reuse_hits=True)
        rows[:, cap] = self.key_counts          # (a row that overflowed has lost its last key to the header: the count says so)
        return rows

    def _k1_stage(self, cb, gx, umi, meta, n, draws):
        """hit count → draw-rank base → probe/filter/pack → count exchange; returns (send, recv) counts on the host"""
        G, st = self.G, self.st
        self._small.zero_()
        if umi is None:
            st.count_hits(cb, n, self.hits, blocked=gx)          # the cell indices go into the blocked buffer's scratch slices
        else:
            st.count_hits(cb, n, self.hits)
        if self.pipelined and self.host_small:                               # gloo: host integers
            all_h = torch.empty(G, dtype=torch.int64)
            self._gather_small(all_h, to_host_tensor(self.hits))
            self.draw_base.fill_(int(all_h[:self.rank].sum()))
            st.probe_pack(cb, gx, umi, meta, n, draws, self.draw_base, self.keys_out, self.stride,
                          self.key_counts, self.counters, reuse_hits=True)
            send = to_host_tensor(self.key_counts)
            recv = torch.empty(G, dtype=torch.int64)
            self._exchange_small(recv, send)
            return send.tolist(), recv.tolist()
        self._all_gather(self.all_hits, self.hits)
        self.draw_base.copy_(self.all_hits[:self.rank].sum().reshape(1))
        st.probe_pack(cb, gx, umi, meta, n, draws, self.draw_base, self.keys_out, self.stride,
                      self.key_counts, self.counters, reuse_hits=True)
        self._all_to_all_single(self.recv_counts, self.key_counts)
        both = torch.cat([self.key_counts, self.recv_counts]).tolist()       # the one host sync of the pass
        return both[:G], both[G:]

    def _gather_small(self, out_cpu, inp_cpu):
        self.n_collectives += 1
        dist.all_gather_into_tensor(out_cpu, inp_cpu, group=self.small_group)

    def _exchange_small(self, out_cpu, inp_cpu):
        self.n_collectives += 1
        dist.all_to_all_single(out_cpu, inp_cpu, group=self.small_group)

    def prepare_draws(self, draws):
        """a resident draw stream in the form the stages read fastest (HIP: the decision bits, made once); pass the result
        to run() in place of `draws`"""
        return self.st.draw_bits(draws) if hasattr(self.st, "draw_bits") else draws

    def _run_wide(self, cb, gx, umi, meta, n, draws, umi_ext):
        """one pass with keys wider than 64 bits (module docstring): exact protocol, the caller's stream, two words per key"""
        G, st = self.G, self.st
        if umi is None:
            raise ValueError("keys wider than 64 bits take SoA records (cb, gx, umi, meta [, umi_ext]), not the blocked layout")
        if getattr(self, "_wide_vals", None) is None:
            self._wide_vals = torch.empty((G, self.stride), dtype=torch.int64, device=self.dev)
            self._wide_recv_v = torch.empty(self.recv_cap, dtype=torch.int64, device=self.dev)
        self._use_slot(0)
        self.recv = self._recv_slots[0]
        self._small.zero_()
        st.count_hits(cb, n, self.hits)
        if G > 1:
            self._all_gather(self.all_hits, self.hits)
            self.draw_base.copy_(self.all_hits[:self.rank].sum().reshape(1))
        st.probe_pack_wide(cb, gx, umi, meta, umi_ext, n, draws, self.draw_base, self.keys_out, self._wide_vals, self.stride,
                           self.key_counts, self.counters, reuse_hits=True)
        if G > 1:
            self._all_to_all_single(self.recv_counts, self.key_counts)
            both = torch.cat([self.key_counts, self.recv_counts]).tolist()
            send, recv = both[:G], both[G:]
            if max(send) > self.stride:
                raise RuntimeError("a destination buffer overflowed (%d keys, room for %d)" % (max(send), self.stride))
            self.n_recv = int(sum(recv))
            pack = lambda buf: torch.cat([buf[g, :send[g]] for g in range(G)]) if sum(send) else self.recv[:0]
            self._all_to_all_single(self.recv[:self.n_recv], pack(self.keys_out), recv, send)
            self._all_to_all_single(self._wide_recv_v[:self.n_recv], pack(self._wide_vals), recv, send)
            keys, vals = self.recv, self._wide_recv_v
        else:
            self.n_recv = int(self.key_counts[0].item())
            keys, vals = self.keys_out.view(-1), self._wide_vals.view(-1)
        self._wide_rows = st.reduce_wide(keys, vals, self.n_recv)
        self._counters_reduced = False
        self._verified = True
        self._fixed_step = False
        self._gathered = True

    def run(self, cb, gx, umi, meta, n, draws, inputs_ready=None, umi_ext=None):
        """One pass over this rank's slice.  `draws` is the job-wide draw stream (resident), or prepare_draws() of it.

        Pipelined mode runs K1 on its own stream so that it overlaps the previous step's sort on the caller's stream;
        it therefore cannot wait for everything the caller has queued.  A caller that REWRITES the input tensors between
        steps must pass `inputs_ready`, an event recorded on its stream after the writes: K1 waits for it.  Without it
        the inputs must be ready when the first step starts and stay untouched afterwards (bench.py: resident inputs)."""
        G, st = self.G, self.st
        self._wide_rows = None
        if getattr(st, "wide", False):
            if inputs_ready is not None:
                inputs_ready.wait()
            return self._run_wide(cb, gx, umi, meta, n, draws, umi_ext)
        self._last_inputs = (cb, gx, umi, meta, n, draws, inputs_ready)       # (a step repeated by ensure_exact() waits for the same event)
        self._fixed_step = False
        if G > 1 and self.fixed and self.cap is not None:
            self._run_fixed(cb, gx, umi, meta, n, draws, inputs_ready)
            return
        if G > 1:
            b = self._step % len(self._small_slots)
            self._step += 1
            self._use_slot(b)
            if self.pipelined:
                main = torch.cuda.current_stream(self.dev)
                k1 = self.k1_stream
                if inputs_ready is not None:
                    k1.wait_event(inputs_ready)          # the caller refilled the inputs on its own stream
                elif self._step == 1:
                    k1.wait_stream(main)                 # the inputs were produced on the caller's stream
                with torch.cuda.stream(k1):
                    if self._slot_free[b] is not None:
                        k1.wait_event(self._slot_free[b])
                    send, recv = self._k1_stage(cb, gx, umi, meta, n, draws)
            else:
                send, recv = self._k1_stage(cb, gx, umi, meta, n, draws)
            self.n_recv = int(sum(recv))
            self.recv = self._recv_slots[b]
            if self.pipelined:
                x = self.x_stream
                with torch.cuda.stream(x):
                    x.wait_stream(k1)                    # this step's shard buffers are complete
                    if self._recv_free[b] is not None:
                        x.wait_event(self._recv_free[b])  # the step that last used this receive buffer is sorted and reduced
                    self._exchange_keys(send, recv)
                    ev = torch.cuda.Event()
                    ev.record(x)
                self._slot_free[b] = ev                   # K1 may write this slot's shard buffers again after the exchange
                main.wait_event(ev)
            else:
                self._exchange_keys(send, recv)
            self.d_n = self._d_n_buf
            self.d_n.fill_(self.n_recv)
            keys = self.recv
            self._counters_reduced = False              # summed over ranks on demand (global_counters), not per step
            if self.fixed and self.cap is None:
                # what this step has learned sizes the fixed-capacity rows of the later ones: the largest count of any
                # (source, destination) pair of the job, a quarter on top (one more collective, once)
                m = torch.tensor([max(max(send), max(recv))], dtype=torch.int64, device=self.dev)
                self._all_reduce(m, op=dist.ReduceOp.MAX)
                slack = int(os.environ.get("FASTF_DIST_CAP_SLACK", "1024"))
                self.cap = max(1, min(int(m.item()) * 5 // 4 + slack, self.stride - 1))
        else:
            self._small.zero_()
            st.probe_pack(cb, gx, umi, meta, n, draws, self.draw_base, self.keys_out, self.stride,
                          self.key_counts, self.counters)
            self.d_n = self.key_counts[:1]              # the device reads the key count where K1b accumulated it
            self.n_recv = n                             # upper bound
            keys = self.keys_out.view(-1)
        # 4. local sort + reduce
        self.sorted = st.sort_reduce(keys, self.tmp, self.d_n, self.n_recv, self.feature, self.cell,
                                     self.count, self.nnz)
        if G > 1 and self.pipelined:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self._recv_free[b] = ev
        self._keys_buf = keys
        self._verified = False
        self._gathered = not hasattr(st, "rows_gather")   # stages that reduce straight into feature/cell/count

    def _run_fixed(self, cb, gx, umi, meta, n, draws, inputs_ready):
        """one step in the fixed-capacity form: two collectives, no host synchronisation"""
        G, st, cap = self.G, self.st, self.cap
        b = self._step % len(self._small_slots)
        self._step += 1
        self._use_slot(b)
        self.recv = self._recv_slots[b]
        recv_rows = self.recv[:G * (cap + 1)].view(G, cap + 1)
        if self.pipelined:
            main = torch.cuda.current_stream(self.dev)
            k1, x = self.k1_stream, self.x_stream
            if inputs_ready is not None:
                k1.wait_event(inputs_ready)
            elif self._step == 1:
                k1.wait_stream(main)
            with torch.cuda.stream(k1):
                if self._slot_free[b] is not None:
                    k1.wait_event(self._slot_free[b])
                rows = self._k1_fixed(cb, gx, umi, meta, n, draws)
            with torch.cuda.stream(x):
                x.wait_stream(k1)
                if self._recv_free[b] is not None:
                    x.wait_event(self._recv_free[b])
                self._all_to_all_single(recv_rows.view(-1), rows.view(-1))
                ev = torch.cuda.Event()
                ev.record(x)
            self._slot_free[b] = ev
            main.wait_event(ev)
        else:
            rows = self._k1_fixed(cb, gx, umi, meta, n, draws)
            self._all_to_all_single(recv_rows.view(-1), rows.view(-1))
        counts = recv_rows[:, cap].contiguous()
        # (the headers this rank RECEIVED: every count of the job is some receiver's header, and ensure_exact() takes the maximum
        #  over the ranks — reading the sender's own rows here as well raced the K1 stage of the step after next, which
        #  rewrites that slot on its own stream: ADVICE r4)
        self._overflow = (counts > cap).any()
        self.d_n = self._d_n_buf
        st.set_regions(counts.clamp(max=cap), G, cap + 1, self.d_n)
        self.n_recv = G * cap                           # the host knows a bound only; the device reads the count at d_n
        self.sorted = st.sort_reduce(self.recv, self.tmp, self.d_n, self.n_recv, self.feature, self.cell, self.count, self.nnz)
        if self.pipelined:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self._recv_free[b] = ev
        self._keys_buf = self.recv
        self._counters_reduced = False
        self._verified = False
        self._fixed_step = True
        self._gathered = not hasattr(st, "rows_gather")

    def _exchange_keys(self, send, recv):
        """the one exchange of the pass: shard buffer g of this rank → rank g; what arrives lands in self.recv"""
        G = self.G
        # one all_to_all_single with split sizes on both backends (zero-length splits are legal there; the list form of
        # all_to_all with empty tensors is not something to meet for the first time on an 8-GPU node): the shard
        # buffers' valid prefixes are packed into one contiguous send buffer first (8 bytes per key, once)
        flat = torch.cat([self.keys_out[g, :send[g]] for g in range(G)]) if sum(send) else self.recv[:0]
        self._all_to_all_single(self.recv[:self.n_recv], flat, recv, send)

    # ---- collectives: direct on RCCL (and on CPU tensors over gloo), staged through the host otherwise ----
    def _all_gather(self, out, inp):
        self.n_collectives += 1
        if self.host_staged:
            o = to_host_tensor(out)
            dist.all_gather_into_tensor(o, to_host_tensor(inp), group=self.group)
            copy_into(out, o)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def _all_to_all_single(self, out, inp, out_splits=None, in_splits=None):
        self.n_collectives += 1
        if self.host_staged:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, to_host_tensor(inp), out_splits, in_splits, group=self.group)
            copy_into(out, o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def _all_reduce(self, t, op=None):
        self.n_collectives += 1
        op = dist.ReduceOp.SUM if op is None else op
        if self.host_staged:
            h = to_host_tensor(t)
            dist.all_reduce(h, op=op, group=self.group)
            copy_into(t, h)
        else:
            dist.all_reduce(t, op=op, group=self.group)

    def ensure_exact(self):
        """If the group-only sort met runs beyond its cap, finish the sort and reduce again (every key is still in
        the buffers, permuted).  Costs one synchronisation; called before results are read.

        COLLECTIVE when G > 1: after a fixed-capacity step the ranks agree on whether a count outgrew its row (one
        all_reduce), so every rank must call it — and with it gather_rows(), local_coo() and global_counters(), which
        call it first — at the same point of the job, whether or not it wants the rows itself.  Only the LAST step can be
        repeated: a caller that runs several steps before reading checks after each one (bench.py reads after every step's
        warm-up and once at the end of identical steps)."""
        if self._verified:
            return
        if self._fixed_step:
            # did a count outgrow the rows (the input changed since the capacity was learned)?  A rank sees the headers of its
            # own rows only, so the ranks agree first (results are being read: a collective and a synchronisation are due
            # anyway); if so, this step again, the exact way, which also learns the new capacity
            o = self._overflow.to(torch.int64).reshape(1)
            self._all_reduce(o, op=dist.ReduceOp.MAX)
            if int(o.item()):
                self.cap = None
                self.run(*self._last_inputs)
        self._verified = True
        st = self.st
        if getattr(st, "run_too_long", None) and st.run_too_long():
            st.skip_low = False                         # this data has deep groups: later passes sort fully right away
            src = self.sorted
            other = self.tmp if src.data_ptr() != self.tmp.data_ptr() else self._keys_buf
            self.sorted = st.sort_reduce(src, other, self.d_n, self.n_recv, self.feature, self.cell,
                                         self.count, self.nnz, fresh=False)
            self._gathered = not hasattr(st, "rows_gather")

    def gather_rows(self):
        """rows of the last pass into self.feature / .cell / .count (device), if the reduce left them segmented"""
        self.ensure_exact()
        if not getattr(self, "_gathered", True):
            self.st.rows_gather(self.d_n, self.feature, self.cell, self.count)
            self._gathered = True

    def local_coo(self):
        if getattr(self, "_wide_rows", None) is not None:
            return self._wide_rows
        self.gather_rows()
        z = int(self.nnz.item())
        return (to_host(self.feature[:z]).astype(np.int64), to_host(self.cell[:z]).astype(np.int64),
                to_host(self.count[:z]).astype(np.int64))

    def gather_coo(self):
        """All shards' rows on every rank, merged into the reference order (cell, feature)."""
        f, c, k = self.local_coo()
        if self.G == 1:
            return f, c, k
        z = torch.tensor([len(f)], dtype=torch.int64, device=self.dev)
        zs = torch.zeros(self.G, dtype=torch.int64, device=self.dev)
        self._all_gather(zs, z)
        zmax = int(zs.max().item())
        pad = torch.zeros((3, zmax), dtype=torch.int64, device=self.dev)
        pad[0, :len(f)] = to_device(f, self.dev)
        pad[1, :len(f)] = to_device(c, self.dev)
        pad[2, :len(f)] = to_device(k, self.dev)
        allp = torch.zeros((self.G, 3, zmax), dtype=torch.int64, device=self.dev)
        self._all_gather(allp.view(-1), pad.view(-1))
        allp, zs = to_host(allp), to_host(zs)
        F = np.concatenate([allp[g, 0, :zs[g]] for g in range(self.G)])
        Cc = np.concatenate([allp[g, 1, :zs[g]] for g in range(self.G)])
        K = np.concatenate([allp[g, 2, :zs[g]] for g in range(self.G)])
        order = np.lexsort((F, Cc))
        return F[order], Cc[order], K[order]

    def global_counters(self):
        """{hits, sampled, sampled_valid, error bits} of the whole job (collective when G > 1: every rank must call it)"""
        self.ensure_exact()
        if self.G > 1 and not self._counters_reduced:
            self._all_reduce(self.counters[:3])
            self._counters_reduced = True
        c = to_host(self.counters)
        return int(c[0]), int(c[1]), int(c[2]), int(c[3])
