"""Software pipeline over consecutive image batches: host images in, host token ids out.

The reference's ``generate`` is strictly one image at a time (caption_models.py:48-74): image -> encoder -> decoder loop ->
ids.  Batched over images this hot path has two very different halves on an MI355X: the ResNet-50 encoder is a
throughput-bound stack of large GEMM-shaped launches, the decode positions are chains of small latency-bound launches
that leave most of the 256 CUs idle.  ``CaptionPipeline`` therefore runs, on separate HIP streams,

    copy stream  : pinned host images of batch i+1  -> HBM           (async H2D, double-buffered)
    encode stream: encoder of batch i+1                               (waits for its copy)
    decode stream: beam-search decode of batch i, ids -> pinned host  (waits for its encoder; async D2H)

so that the copy and the encoder of the next batch fill the CUs / the PCIe link while the current batch decodes.
Captions are the ones ``generate_batch`` gives for the same ``seed`` / ``img0`` (the streams only reorder independent
work).  Nothing here computes: every operation is a kernel of libdeephumor_hip.so or a copy.
"""
import torch

from . import hip

__all__ = ["CaptionPipeline", "u8_preprocess"]


def u8_preprocess(model, size=(224, 224)):
    """Input stage for DECODED images: ``uint8 [N, H, W, 3]`` (what an image decoder produces; 38.5 MB per 256 images at
    224 x 224 instead of the 154 MB of the fp32 NCHW batch) -> what ``model.encode`` takes.  The notebook's transform
    (Resize -> ToTensor -> Normalize, deephumor_demo.ipynb:565-567) runs on the device: for a 16-bit model the
    normalised batch is written directly in the packed channels-last layout the stem kernel reads
    (``dh_normalize_pack_u8``; no fp32 tensor exists), for the fp32 model it is the fp32 NCHW batch the reference's
    encoders take.  Further inputs (labels) pass through."""
    from .experiments.inference import preprocess_images
    dtype = next(model.parameters()).dtype

    def stage(images_u8, *rest):
        return (preprocess_images(images_u8, size=size, dtype=dtype),) + rest
    return stage


class CaptionPipeline:
    def __init__(self, model, overlap=True, preprocess=None, **gen_kw):
        """``preprocess``: optional callable mapping the staged device tensors of a batch to ``model.encode``'s inputs
        (e.g. ``u8_preprocess(model)``); it runs on the encode stream in front of the encoder."""
        self.model, self.gen_kw, self.overlap, self.preprocess = model, gen_kw, overlap, preprocess
        self.dev = next(model.parameters()).device
        if overlap:
            # (a high-priority decode stream was measured in round 4: no gain, the encoder's workgroups hold the CUs until they retire
            #  whatever the queue priority)
            self.copy_s, self.enc_s, self.dec_s = (torch.cuda.Stream(device=self.dev) for _ in range(3))
        else:
            self.copy_s = self.enc_s = self.dec_s = torch.cuda.current_stream(self.dev)
        # staging buffers, ONE per (slot, input) / output slot: a batch of another shape replaces the slot's buffer (a service fed
        # images of varying sizes would otherwise keep a device buffer per shape it has ever seen)
        self._dev_in = {}            # (slot, input index) -> device staging tensor (double-buffered)
        self._host_out = {}          # output slot -> pinned (tokens, lengths) pair, three slots used in turn
        self._slot = 0
        self._out_slot = 0
        self._slot_reader = {}       # staging slot -> event recorded after the encoder that read it
        # decode is queued without a host read of the beam engine's error word (examined at hand-over); decoders that cannot defer it
        # (pad_index == 1: host-driven full re-forward) decode synchronously
        self._async = getattr(getattr(model, "decoder", None), "pad_index", 0) != 1

    # -- stages ------------------------------------------------------------------------------------------------------
    def _stage(self, host_inputs):
        """H2D of one batch (tensors in pinned host memory copy asynchronously) on the copy stream; device-resident
        tensors pass through."""
        slot, out = self._slot, []
        self._slot ^= 1
        self._last_slot = slot
        if slot in self._slot_reader:                 # two batches may be in flight: the slot's previous reader must be done
            self.copy_s.wait_event(self._slot_reader[slot])
        producer = torch.cuda.current_stream(self.dev)
        if self.overlap and any(t.is_cuda for t in host_inputs):
            # a device-resident input may still be being written on the caller's stream (e.g. by preprocess_images): the
            # event recorded below must not fire before that work has finished
            self.copy_s.wait_stream(producer)
        with torch.cuda.stream(self.copy_s):
            for j, t in enumerate(host_inputs):
                if t.is_cuda:
                    if self.overlap:
                        t.record_stream(self.enc_s)
                    out.append(t)
                    continue
                key = (slot, j)
                buf = self._dev_in.get(key)
                if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                    if buf is not None and self.overlap:          # the old buffer may still be read on the side streams
                        buf.record_stream(self.copy_s)
                        buf.record_stream(self.enc_s)
                    buf = self._dev_in[key] = torch.empty(t.shape, dtype=t.dtype, device=self.dev)
                buf.copy_(t, non_blocking=True)
                out.append(buf)
            ev = torch.cuda.Event()
            ev.record(self.copy_s)
        return out, ev

    def _encode(self, staged, ev):
        with torch.cuda.stream(self.enc_s), torch.no_grad():
            self.enc_s.wait_event(ev)
            if self.preprocess is not None:
                staged = self.preprocess(*staged)
            enc = self.model.encode(*staged)
            done = torch.cuda.Event()
            done.record(self.enc_s)
            self._slot_reader[self._last_slot] = done
        for t in enc:
            t.record_stream(self.dec_s)
        return enc, done

    def _decode(self, enc, ev, seed, img0, to_host):
        """Queues the decode of one batch on the decode stream and returns WITHOUT waiting for it: the beam engine's error word is
        not read here (``defer_check``) but copied to pinned memory behind the token ids and examined by ``_finish`` once the
        batch's event has fired."""
        from .models.beam import resolve_seed
        if self.gen_kw.get("rng") != "torch":
            seed = resolve_seed(seed)                 # fixed now: a repeated decode (BeamOverflow) must draw the same noise
        with torch.cuda.stream(self.dec_s), torch.no_grad():
            self.dec_s.wait_event(ev)
            res = self.model.decode(enc, seed=seed, img0=img0, defer_check=self._async, **self.gen_kw)
            toks, lens = res[0], res[1]
            err_host = None
            if len(res) > 2:
                err_host = torch.empty((1,), dtype=torch.int32).pin_memory()
                err_host.copy_(res[2].view(-1)[:1], non_blocking=True)
            slot = None
            if to_host:
                slot = self._out_slot
                self._out_slot = (self._out_slot + 1) % 3
                toks, lens = self._to_host(toks, lens, slot)
            done = torch.cuda.Event()
            done.record(self.dec_s)
        return dict(toks=toks, lens=lens, done=done, err=err_host, redo=(enc, seed, img0, to_host, slot))

    def _to_host(self, toks, lens, key):
        bufs = self._host_out.get(key)
        if bufs is None or bufs[0].shape != toks.shape:    # (a replaced pinned pair stays alive while a consumer holds it)
            bufs = self._host_out[key] = (torch.empty(toks.shape, dtype=toks.dtype).pin_memory(),
                                          torch.empty(lens.shape, dtype=lens.dtype).pin_memory())
        bufs[0].copy_(toks, non_blocking=True)
        bufs[1].copy_(lens, non_blocking=True)
        return bufs

    def _finish(self, q):
        """Waits for a queued batch and returns its ``(tokens, lengths)``; reads the deferred error word: flat logits that
        overflowed the pre-filtered samplers (``BeamOverflow``) repeat this batch through the general sampler, anything else raises
        as ``generate_batch`` does."""
        from .models.beam import BeamOverflow, BeamSearchHelper, warn_overflow_retry
        q["done"].synchronize()
        if q["err"] is not None:
            try:
                BeamSearchHelper.raise_for(int(q["err"][0]))
            except BeamOverflow:
                warn_overflow_retry()
                enc, seed, img0, to_host, slot = q["redo"]
                with torch.cuda.stream(self.dec_s), torch.no_grad():
                    toks, lens = self.model.decode(enc, seed=seed, img0=img0, exact=True, **self.gen_kw)
                    if to_host:           # into the pinned pair THIS batch already owns (the other two may still be held by the consumer)
                        toks, lens = self._to_host(toks, lens, slot)
                self.dec_s.synchronize()
                return toks, lens
        return q["toks"], q["lens"]

    # -- driver ------------------------------------------------------------------------------------------------------
    def run(self, batches, seeds=None, img0=0, to_host=True, low_latency=False):
        """``batches``: iterable of input tuples (``(images,)`` or ``(images, labels)``), host (ideally pinned) or device
        tensors.  Yields ``(tokens, lengths)`` per batch, in order; with ``to_host`` they live in one of THREE pinned host
        buffer pairs used in turn, so a result may be kept across ONE further iteration and ``list(pipe.run(...))`` must clone.
        All work of a batch has completed when it is yielded.

        Decoding is queued asynchronously (the beam engine's error word is examined only when the batch is handed over), in one of
        two schedules:

        * default (throughput): batch i is handed over while batch i+1 decodes -- its decode loop is queued right behind batch i's,
          with no host round trip in between -- and the copy + encoder of batch i+2 are already issued; the hand-over of batch i
          therefore happens after batch i+2 has been FETCHED from ``batches``;
        * ``low_latency=True`` (a service whose iterator blocks until the next request arrives): batch i is handed over as soon as
          it has finished, before anything further is fetched; only the copy + encoder of the already-fetched batch i+1 overlap its
          decode."""
        it = iter(batches)
        seeds = iter(seeds) if seeds is not None else None
        cur = next(it, None)
        if cur is None:
            return
        enc, ev = self._encode(*self._stage(cur))
        if low_latency:
            nxt = next(it, None)
            while cur is not None:
                if nxt is not None:
                    nxt_enc, nxt_ev = self._encode(*self._stage(nxt))
                queued = self._decode(enc, ev, next(seeds) if seeds is not None else None, img0, to_host)
                yield self._finish(queued)            # nothing finished is withheld while the iterator blocks
                cur = nxt
                if nxt is not None:
                    enc, ev = nxt_enc, nxt_ev
                    nxt = next(it, None)
            return
        pending = None
        while cur is not None:
            nxt = next(it, None)
            if nxt is not None:                      # issue the next batch's copy + encoder BEFORE this batch's decode
                nxt_enc, nxt_ev = self._encode(*self._stage(nxt))
            seed = next(seeds) if seeds is not None else None
            queued = self._decode(enc, ev, seed, img0, to_host)          # asynchronous: returns once the launches are queued
            if pending is not None:                  # the PREVIOUS batch is handed over while this one decodes
                yield self._finish(pending)
            pending = queued
            cur = nxt
            if nxt is not None:
                enc, ev = nxt_enc, nxt_ev
        if pending is not None:
            yield self._finish(pending)
