"""Video -> first token pipeline: CPU frame production overlapped with GPU ViT + group prefill.

Reference shape (lvu/models/qwen25_lvu_interleaved.py:237-342, 733-942): a daemon thread pulls frame groups from
the reader, runs the HF processor on the CPU and feeds a bounded Queue(3); the main thread polls it every 10 ms and
runs the group loop.  Here the producer thread only moves uint8 frames into a pinned host ring and enqueues an
async H2D copy on a dedicated HIP stream (event-signalled, no polling); normalise + patchify + ViT run on the GPU on a
second stream, one group ahead of the LLM prefill that runs on the main stream.  Token ids and M-RoPE positions are
built before any pixel exists from (nframes, H, W) alone, like the reference's dummy_call (interleaved:522-638, 786-810).
"""
from __future__ import annotations

import os
import queue
import threading
import time
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import planner
from .engine import QuickPrefillEngine
from .decode import GraphDecoder
from .frames import open_video, smart_nframes
from .native import host_memcpy
from .lvu_config import LVUConfig, effective_k
from .sampling import TokenSelector
from .spec import TextSpec
from .vit import VisionTower, VisionWeights, patchify_frames
from .weights import DecoderWeights


@dataclass
class QwenVLNative:
    """The `model` object of the native plugin: decoder + vision weights resident on one device."""
    text: DecoderWeights
    vision: VisionWeights
    device: torch.device
    name: str = "synthetic"
    rope_deltas: Optional[int] = None
    engine: Optional[QuickPrefillEngine] = None
    config: Optional[LVUConfig] = None
    generation_defaults: Optional[dict] = None      # the checkpoint's generation_config.json (HF generate applies it implicitly)

    @property
    def spec(self) -> TextSpec:
        return self.text.spec


@dataclass
class Timings:
    fetch: float = 0.0          # time the consumer waited for frames (video_processing_time analogue)
    vit: float = 0.0            # device time of patchify + ViT (events)
    prefill: float = 0.0        # group loop wall time, device-synchronised at the end (total_prefill)
    decode: float = 0.0
    e2e: float = 0.0
    ttft: float = 0.0           # first frame requested -> first token id on the host
    tokens: int = 0
    groups: int = 0


class _GpuProgress(threading.Thread):
    """QP_PIPELINE_DEBUG=1: every 5 s, how many groups the host has enqueued and how many ViT passes / group prefills the GPU has
    finished (event queries), on stderr — tells a slow device from a stuck one."""

    def __init__(self, n_groups):
        super().__init__(daemon=True)
        self.n, self.vit, self.pre, self.halt, self.t0 = n_groups, [], [], threading.Event(), time.perf_counter()
        self.start()

    def enqueued(self, vit_done, prefill_done):
        self.vit.append(vit_done); self.pre.append(prefill_done)

    def run(self):
        import sys
        while not self.halt.wait(5.0):
            nv, npf = sum(e.query() for e in list(self.vit)), sum(e.query() for e in list(self.pre))
            print(f"[pipeline {time.perf_counter() - self.t0:6.1f}s] enqueued {len(self.pre)}/{self.n} groups; finished on the GPU: "
                  f"{nv} ViT passes, {npf} group prefills", file=sys.stderr, flush=True)

    def stop(self):
        self.halt.set()


class _Producer(threading.Thread):
    """Frame groups -> pinned ring -> async H2D on `copy_stream`; bounded like the reference's Queue(maxsize=3)."""

    def __init__(self, reader, n_groups, frames_per_group, device, depth=3, ring_cache=None):
        super().__init__(daemon=True)
        self.reader, self.n_groups, self.device, self.depth = reader, n_groups, device, depth
        # pinned host slots + device slots are kept across videos by the owner of `ring_cache` (page-locking ~80 MB costs tens
        # of ms: it would otherwise sit in front of the first group of every video)
        self.ring_cache = ring_cache if ring_cache is not None else {}
        self.q: "queue.Queue" = queue.Queue(maxsize=depth)
        self.exc = None
        self.use_gpu = device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(device) if self.use_gpu else None
        self.slots_free = threading.Semaphore(depth)
        self.ring = None
        self.fpg = frames_per_group
        self.h2d_done = [None] * depth      # per slot: event after the last H2D copy out of the pinned buffer (copy stream)
        self.read_done = [None] * depth     # per slot: event after the consumer's last GPU read of the device buffer (ViT stream)

    def run(self):
        try:
            for g in range(self.n_groups):
                frames = next(self.reader)                      # uint8 [g,3,H,W] (CPU work, GIL released inside numpy)
                self.slots_free.acquire()
                if self.use_gpu:
                    if self.ring is None:
                        shape = (self.fpg,) + tuple(frames.shape[1:])
                        key = (self.depth, shape, str(self.device))
                        if key not in self.ring_cache:
                            self.ring_cache.clear()                 # one video geometry at a time
                            self.ring_cache[key] = [(torch.empty(shape, dtype=torch.uint8).pin_memory(),
                                                     torch.empty(shape, dtype=torch.uint8, device=self.device)) for _ in range(self.depth)]
                        self.ring = self.ring_cache[key]
                    slot = g % self.depth
                    host, dev = self.ring[slot]
                    # slot reuse is ordered explicitly, not by luck: (1) the pinned host buffer may only be overwritten once the
                    # previous H2D copy out of it has finished (host-side wait, this thread only); (2) the device buffer may only
                    # be overwritten once the consumer's last GPU read of it (patchify on the ViT stream) has finished — the
                    # consumer hands that event back through release() and the copy stream waits on it.
                    if self.h2d_done[slot] is not None:
                        self.h2d_done[slot].synchronize()
                    # native, GIL-free fill of the pinned slot (qp_host_memcpy through ctypes; 2-10x faster than Tensor.copy_ here,
                    # whose speed follows torch's process-wide intra-op thread count): the launching thread is never starved
                    host_memcpy(host[: frames.shape[0]], frames.contiguous())
                    with torch.cuda.stream(self.copy_stream):
                        if self.read_done[slot] is not None:
                            self.copy_stream.wait_event(self.read_done[slot])
                        dev[: frames.shape[0]].copy_(host[: frames.shape[0]], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self.copy_stream)
                    self.h2d_done[slot] = ev
                    self.q.put((g, dev[: frames.shape[0]], ev))
                else:
                    self.q.put((g, frames.clone(), None))
        except BaseException as e:   # re-raised in the consumer, like interleaved:291-292, 314-316
            self.exc = e
            self.q.put(None)

    def get(self):
        item = self.q.get()
        if item is None:
            raise self.exc
        return item

    def release(self, g: int = 0, read_done=None):
        """Group g's slot may be refilled; `read_done` = event recorded after the last GPU read of its device buffer."""
        self.read_done[g % self.depth] = read_done
        self.slots_free.release()


class PrefillPipeline:
    def __init__(self, model: QwenVLNative, config: LVUConfig, processor, ops=None):
        self.model, self.cfg, self.processor, self.ops = model, config, processor, ops
        self.use_gpu = model.device.type == "cuda"
        self._tower = None
        self.vit_stream = torch.cuda.Stream(model.device) if self.use_gpu else None
        self.last_timings: Optional[Timings] = None

    @property
    def tower(self) -> VisionTower:
        if self._tower is None:
            ops = self.ops
            if ops is None and self.use_gpu:
                from .native import QuickPrefillOps
                ops = self.ops = QuickPrefillOps(self.model.device)
            self._tower = VisionTower(self.model.vision, ops=ops if self.use_gpu else None)
        return self._tower

    # ------------------------------------------------------------------ planning (no pixels needed)
    def plan(self, reader, question: str):
        cfg, spec = self.cfg, self.model.spec
        total, vfps = len(reader), reader.get_fps()
        nframes = smart_nframes(total, vfps, nframes=cfg.num_frames if cfg.fps is None else None, fps=cfg.fps)
        ek = cfg.extra_kwargs or {}
        src_h = getattr(reader, "src_h", None) or reader.height
        src_w = getattr(reader, "src_w", None) or reader.width
        if src_h is None:                                            # array-backed video: frames already at model size
            src_h, src_w = reader.arr.shape[2:]
            H, W = src_h, src_w
        else:
            H, W = planner.video_frame_size(nframes, src_h, src_w, ek.get("max_pixels"), ek.get("min_pixels"))
        idx = np.linspace(0, total - 1, nframes).round().astype(np.int64)   # interleaved:397-399
        vs = self.model.vision.spec
        gh, gw = H // vs.patch_size, W // vs.patch_size
        prompt = self.processor.build_prompt(question)
        n_video = (nframes // vs.temporal_patch_size) * (gh // 2) * (gw // 2)
        T = len(prompt.prefix_ids) + n_video + len(prompt.tail_ids)
        gs = cfg.video_group_size
        plan = planner.plan_groups(nframes, gs, gh, gw, len(prompt.prefix_ids), T, vs.temporal_patch_size, vs.spatial_merge_size)
        # Qwen2.5-VL: tokens_per_second * second_per_grid_t with second_per_grid_t = temporal_patch / sampled fps
        sample_fps = nframes / max(total / vfps, 1e-9)             # qwen-vl-utils: video_sample_fps = nframes / total_frames * video_fps
        tscale = spec.resolved_temporal_scale(sample_fps, vs.temporal_patch_size)
        pos, delta = planner.mrope_positions(len(prompt.prefix_ids), (nframes // vs.temporal_patch_size, gh, gw), len(prompt.tail_ids),
                                             vs.spatial_merge_size, tscale)
        return dict(nframes=nframes, H=H, W=W, idx=idx, prompt=prompt, plan=plan, pos=pos, delta=delta, T=T, gh=gh, gw=gw)

    def _engine(self, plan, T, max_new_tokens: int = 0) -> QuickPrefillEngine:
        cfg, spec = self.cfg, self.model.spec
        kept = sum((effective_k(n, cfg, 0, spec.n_layers) or n) for n in plan.tokens)
        need_cap = kept + plan.tail_len + max(256, max_new_tokens + 8)        # room for every token that will be decoded
        n_max = max(plan.tokens + [plan.tail_len, 1])
        if cfg.query_based:                                   # the prompt tokens ride along with every group (qwen25_lvu.py:684-686)
            n_max += plan.tail_len
        eng = self.model.engine
        if eng is None or eng.arena.capacity < need_cap or eng.n_max < n_max or eng.cfg is not cfg:
            eng = QuickPrefillEngine(self.model.text, cfg, capacity=need_cap, max_group_tokens=n_max, device=self.model.device, ops=self.ops)
            self.model.engine = eng
        eng.reset()
        return eng

    # ------------------------------------------------------------------ video -> tokens
    @torch.no_grad()
    def generate(self, question: str, video, max_new_tokens: int = 16, overlap: bool = True, eos_token_id: Optional[int] = None,
                 do_sample: Optional[bool] = None, temperature: Optional[float] = None, top_k: Optional[int] = None,
                 top_p: Optional[float] = None, repetition_penalty: Optional[float] = None, seed: Optional[int] = None,
                 num_beams: int = 1, **unused) -> List[int]:
        """generation kwargs as the reference hands them to HF `generate` (qwen25_lvu.py:744-761); unset ones fall back to the
        checkpoint's generation_config.json (`model.generation_defaults`), then to greedy.  Beam search is refused, not ignored."""
        if num_beams != 1:
            raise NotImplementedError("beam search is not implemented (greedy and sampling with temperature / top-k / top-p / repetition penalty are)")
        gd = getattr(self.model, "generation_defaults", None) or {}
        pick = lambda v, k: gd.get(k) if v is None else v
        selector = TokenSelector(pick(do_sample, "do_sample") or False, pick(temperature, "temperature"), pick(top_k, "top_k"),
                                 pick(top_p, "top_p"), pick(repetition_penalty, "repetition_penalty"), seed, device=self.model.device)
        tm = Timings()
        dev = self.model.device
        t_e2e = time.perf_counter()
        reader = open_video(video)
        P = self.plan(reader, question)
        plan, pos = P["plan"], torch.from_numpy(P["pos"]).to(dev)
        reader.height, reader.width, reader.interpolation = P["H"], P["W"], "LANCZOS"
        gs = plan.frames[0]
        reader.frame_iter = gs
        reader.process(P["idx"])                                      # decoding starts here (interleaved:438-442)
        eng = self._engine(plan, P["T"], max_new_tokens)
        self.model.rope_deltas = P["delta"]                           # qwen25_lvu.py:620
        prefix = torch.tensor(P["prompt"].prefix_ids, dtype=torch.long, device=dev)
        tail = torch.tensor(P["prompt"].tail_ids, dtype=torch.long, device=dev)
        # overlapped: bounded ring of 3 groups like the reference's Queue(maxsize=3); sequential: everything is fetched first
        if not hasattr(self, "_ring_cache"):
            self._ring_cache = {}
        prod = _Producer(reader, len(plan.tokens), gs, dev, depth=3 if overlap else len(plan.tokens), ring_cache=self._ring_cache)
        if overlap:
            prod.start()
        else:
            prod.run()                                                # sequential plugin: fetch everything first
        sync = (lambda: torch.cuda.synchronize(dev)) if self.use_gpu else (lambda: None)

        def vit_group(g):
            t0 = time.perf_counter()
            gi, frames, ev = prod.get()
            tm.fetch += time.perf_counter() - t0
            if self.use_gpu:
                s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.vit_stream.wait_event(ev)
                with torch.cuda.stream(self.vit_stream):
                    s_ev.record(self.vit_stream)
                    rows, grid = patchify_frames(frames, self.model.vision.spec, self.model.text.embed.dtype)
                    read_done = torch.cuda.Event()
                    read_done.record(self.vit_stream)             # last read of the ring's device slot
                    feats = self.tower.forward(rows, grid)
                    e_ev.record(self.vit_stream)
                return feats, (s_ev, e_ev), read_done
            rows, grid = patchify_frames(frames, self.model.vision.spec, self.model.text.embed.dtype)
            return self.tower.forward(rows, grid), None, None

        t_pre = time.perf_counter()
        start, vit_events = 0, []
        dbg = _GpuProgress(len(plan.tokens)) if (self.use_gpu and os.environ.get("QP_PIPELINE_DEBUG")) else None
        # query-based predict types: the prompt (everything after the last video token) is appended to every group and scores its
        # keys (qwen25_lvu.py:661-664, 684-689); positions are then the group's AND the next tail_len of the sequence
        q_m = plan.tail_len if (self.cfg.query_based and self.cfg.enable) else 0
        tail_emb = eng.embed_tokens(tail) if q_m else None
        nxt = vit_group(0)
        for g, n in enumerate(plan.tokens):
            feats, evs, read_done = nxt
            if self.use_gpu:
                main = torch.cuda.current_stream(dev)
                main.wait_event(evs[1])
                # feats was allocated on the ViT stream and is read on the main stream (cat / copy into the engine's buffer): tell
                # the caching allocator, or ViT(g+2) could be handed the same block while prefill(g) is still queued
                feats.record_stream(main)
                vit_events.append(evs)
            emb = torch.cat([eng.embed_tokens(prefix), feats], 0) if g == 0 else feats
            assert emb.shape[0] == n, (emb.shape, n)
            if g + 1 < len(plan.tokens):
                nxt = vit_group(g + 1)                                # ViT of the next group runs ahead on its own stream
            eng.prefill_group(emb, pos[:, start:start + n + q_m], prompt_embeds=tail_emb)
            prod.release(g, read_done)
            start += n
            if dbg is not None:
                dbg.enqueued(evs[1], torch.cuda.current_stream(dev).record_event())
        sync()
        if dbg is not None:
            dbg.stop()
        tm.prefill = time.perf_counter() - t_pre
        tm.tokens, tm.groups = start, len(plan.tokens)
        t_dec = time.perf_counter()
        logits = eng.prefill_tail(eng.embed_tokens(tail), pos[:, start:])     # pruning off for the tail (qwen25_lvu.py:737-742)
        if not selector.trivial:                                               # HF processors see the whole prompt (video pads included)
            selector.observe(list(P["prompt"].prefix_ids) + [self.model.spec.video_token_id] + list(P["prompt"].tail_ids),
                             logits.shape[-1], dev)
        tok = selector.select(logits) if not selector.trivial else int(torch.argmax(logits).item())   # first token on the host = TTFT point
        tm.ttft = time.perf_counter() - t_e2e
        out = [tok]
        graph = GraphDecoder.supported(eng)                                   # one hipGraph replay per token (decode.py)
        if graph and getattr(eng, "_graph_decoder", None) is None:
            eng._graph_decoder = GraphDecoder(eng)
        if graph and selector.trivial:                                        # argmax stays on the device inside the graph
            out += eng._graph_decoder.generate(tok, max_new_tokens - 1, P["delta"], eos_token_id)
        else:                                                                 # logits come back per step: processors / sampling, or the
            if graph and max_new_tokens > 1:                                  # per-op path (CPU test doubles, parallel engines)
                eng._graph_decoder.begin(P["delta"])
            for _ in range(max_new_tokens - 1):
                if eos_token_id is not None and tok == eos_token_id:
                    break
                if graph:
                    logits = eng._graph_decoder.step(tok)
                else:
                    logits = eng.decode_step(eng.embed_tokens(torch.tensor([tok], device=dev)), P["delta"])
                tok = selector.select(logits)
                out.append(tok)
        sync()
        tm.decode = time.perf_counter() - t_dec
        tm.e2e = time.perf_counter() - t_e2e
        if self.use_gpu:
            tm.vit = sum(s.elapsed_time(e) for s, e in vit_events) * 1e-3
        self.last_timings = tm
        return out
