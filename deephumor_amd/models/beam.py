"""Batched, device-resident beam-search bookkeeping.

Mirror of ``deephumor.models.beam.BeamSearchHelper`` (reference beam.py:4-112): same constructor
arguments and the same selection rules, but the state of ALL images of a batch lives in HBM and
every step is two kernel launches (``dh_beam_row_sample`` + ``dh_beam_select``; opt-in one: ``dh_beam_step_groups``) with no host
synchronisation, instead of per-image torch ops and a ``torch.all`` sync per token.

Randomness: the reference draws with ``torch.multinomial`` from the global CPU generator.  On
CPU that is exactly "top-k of p / Exp(1) noise" (SURVEY.md section 7), so the kernels take the
noise either from a counter-based Philox stream keyed by ``(seed, global image index, step, row,
token)`` -- results do not depend on batch composition or rank layout -- or, for RNG-replay parity
tests, from caller-supplied tensors (``noise_source``).
"""
import os

import torch

from .. import hip


class BeamOverflow(RuntimeError):
    """A row had more logits at its top-k threshold than the pre-filtered samplers' candidate buffers hold (flat / constant logits):
    the decoders catch this and repeat the batch with ``exact=True`` (the general sampler, which draws such rows over the whole row)."""


class _TorchCpuStream:
    """Replica of the stream a seeded ``torch.Generator`` (CPU) feeds ``Tensor.exponential_`` with, generated with numpy so that a
    batch's images can be filled on a thread pool (torch fills serially under the GIL, ~27 ns per value: 256 images x 5 rows x
    36,541 values take 1.2 s per decode step).  What the pinned torch (2.x CPU, ``exponential_kernel_default``) does per element:
    two mt19937 draws -> ``random64`` (first draw = high word) -> ``u = (x & (2^53 - 1)) * 2^-53`` -> ``-log1p(-u)`` in double ->
    cast to the tensor's type (ATen/core/TransformationHelper.h:144, DistributionTemplates.h); ``manual_seed(s)`` is mt19937's
    standard ``init_genrand(s & 0xffffffff)``.  numpy's ``log1p`` is not glibc's (1 ulp of a double apart on ~7 % of the inputs),
    which would change the float32 result when the double lies within a few ulps of a float32 rounding boundary (~2^-26 of the
    samples): exactly those samples are recomputed with ``math.log1p`` (the C library's).  ``self_check`` compares a long fill
    against torch itself; ``TorchRngNoise`` falls back to torch's own fill if it ever fails (another torch build)."""

    _ok = None

    def __init__(self, seed):
        import numpy as np
        key = np.empty(624, dtype=np.uint64)
        s = int(seed) & 0xFFFFFFFF
        key[0] = s
        for j in range(1, 624):
            s = (1812433253 * (s ^ (s >> 30)) + j) & 0xFFFFFFFF
            key[j] = s
        self.bg = np.random.MT19937()
        self.bg.state = {"bit_generator": "MT19937", "state": {"key": key.astype(np.uint32), "pos": 624}}

    def exponential(self, n):
        """The next ``n`` values of ``torch.empty(n).exponential_(1, generator=g)`` as a float32 numpy array."""
        import math
        import numpy as np
        raw = self.bg.random_raw(2 * n)
        x = ((raw[0::2] << np.uint64(32)) | raw[1::2]) & np.uint64((1 << 53) - 1)
        u = x.astype(np.float64) * (2.0 ** -53)
        y = -np.log1p(-u)
        out = y.astype(np.float32)
        frac = (y.view(np.uint64) & np.uint64(0x1FFFFFFF)).astype(np.int64)      # the 29 bits float32 rounds away
        for i in np.nonzero(np.abs(frac - 0x10000000) <= 16)[0].tolist():
            out[i] = np.float32(-math.log1p(-float(u[i])))
        return out

    @classmethod
    def self_check(cls):
        if cls._ok is None:
            g = torch.Generator().manual_seed(20240229)
            mine = cls(20240229)
            cls._ok = all(bool((torch.empty(n).exponential_(1, generator=g).numpy() == mine.exponential(n)).all())
                          for n in (7, 200000, 36541))
        return cls._ok


class TorchRngNoise:
    """``rng="torch"``: the Exp(1) noise of every draw taken from torch CPU generators in the reference's order and shapes, so that
    the sampled caption is the one the reference returns under the same generator state.  ``torch.multinomial(p, k)`` on the CPU
    is ``topk(p / empty_like(p).exponential_(1), k)``: the reference consumes, per decode step, one ``[n_rows, V]`` fill for the
    row draw (beam.py:39-48 via :57-58; one row at the first step), one ``[n_candidates]`` fill for the candidate draw
    (rnn_models.py:116-121, transformers.py:557-562; ``n_candidates = sum(1 if ended else beam)``, beam.py:72-101) and one
    ``[beam]`` fill for the final draw (rnn_models.py:140, transformers.py:576); nothing once every beam has ended (the ``break`` at
    rnn_models.py:131 / transformers.py:585).

    ``seed=None`` (one image only): the draws come from torch's DEFAULT generator -- ``torch.manual_seed(s); model.generate(image,
    rng="torch")`` is then the reference's own call sequence.  ``seed=int``: image ``i`` of the batch (global index ``img0 + i``)
    draws from its own ``torch.Generator().manual_seed(seed + img0 + i)`` -- row ``i`` equals the reference's
    ``torch.manual_seed(seed + img0 + i); model.generate(image_i)``, whatever the batch composition or rank layout.

    A parity feature, not a throughput path: the noise is generated on the host (rows x V floats per step) and the ended flags are
    read back once per step."""

    def __init__(self, seed, n_img, img0=0, state0=None):
        """``state0``: for ``seed=None``, the default generator's state at the start of the ``generate`` call (``torch.get_rng_state()``
        taken ONCE by the caller): a repeated session (``BeamOverflow`` retry builds a new noise source) then replays the same
        draws instead of continuing from wherever the first attempt left the generator."""
        if seed is None and n_img != 1:
            raise ValueError('rng="torch" with seed=None draws from torch\'s default generator, which only one image at a time can '
                             'consume in the reference\'s order: pass seed=<int> (image i then draws from manual_seed(seed + i))')
        self.seed, self.n_img, self.img0 = seed, int(n_img), int(img0)
        self.gens, self.h, self._state0 = None, None, state0

    def attach(self, helper):
        """Called by the helper that will consume the noise; (re-)positions the generators at the start of their streams, so a
        repeated session (``BeamOverflow`` retry) replays the same draws."""
        self.h = helper
        if self.seed is None:
            if self._state0 is None:
                self._state0 = torch.get_rng_state()
            else:
                torch.set_rng_state(self._state0)
        elif self.n_img >= 8 and _TorchCpuStream.self_check():
            self.gens = [_TorchCpuStream(int(self.seed) + self.img0 + i) for i in range(self.n_img)]     # thread-parallel fills
        else:
            self.gens = [torch.Generator().manual_seed(int(self.seed) + self.img0 + i) for i in range(self.n_img)]

    def _exp(self, i, n):
        g = None if self.gens is None else self.gens[i]
        if isinstance(g, _TorchCpuStream):
            return torch.from_numpy(g.exponential(n))
        return torch.empty(n).exponential_(1, generator=g)

    def _each_image(self, fn, todo):
        """``fn(i)`` for every image of ``todo``: the images' generators are independent, so a batch fills its [rows, V] noise on a
        thread pool (``exponential_`` releases the GIL; one generator is only ever touched by one task at a time)."""
        if self.gens is None or len(todo) < 8:
            for i in todo:
                fn(i)
            return
        pool = self.__dict__.get("_pool")
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self._pool = ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 8) - 1)))
        list(pool.map(fn, todo))

    def __call__(self, kind, step, shape):
        h, b = self.h, self.h.beam_size
        if kind == "multinomial":             # the method surface (sample_k_indices): one fill of the argument's shape
            return self._exp(0, int(torch.Size(shape).numel())).view(shape)
        done = h.done.cpu().tolist()
        out = torch.ones(shape)
        if kind == "row":
            rpi, v = shape[0] // self.n_img, shape[1]

            def fill(i):
                out[i * rpi:(i + 1) * rpi] = self._exp(i, rpi * v).view(rpi, v)
            self._each_image(fill, [i for i in range(self.n_img) if not done[i]])
        elif kind == "cand":
            ended = h._ended.cpu().view(self.n_img, b).tolist()
            for i in range(self.n_img):
                if not done[i]:
                    n_cand = sum(1 if e else b for e in ended[i])
                    out[i, :n_cand] = self._exp(i, n_cand)
        else:                                 # "final": one fill over the beams
            for i in range(self.n_img):
                out[i] = self._exp(i, shape[1])
        return out


class BeamSearchHelper:
    """Beam state for ``n_img`` images x ``beam_size`` beams.

    Reference signature kept: ``BeamSearchHelper(temperature, beam_size, top_k, unk_index,
    eos_index, device)`` (beam.py:7-8); ``n_img``, ``max_len`` and the rest are new keyword arguments.
    """

    def __init__(self, temperature=1.0, beam_size=10, top_k=50, unk_index=1, eos_index=3, device='cuda',
                 n_img=1, max_len=25, src_len=0, seed=0, img0=0, noise_source=None, seed_tensor=None, exact=False):
        assert beam_size <= top_k, '`beam_size` should be less than `top_k`'          # beam.py:9
        self.exact = bool(exact)          # row draws through the general sampler only (see BeamOverflow)
        if beam_size > hip.MAX_BEAMS:     # one wave draws among an image's beams (dh_beam_finalize); the reference has no limit
            raise ValueError(f"beam_size <= {hip.MAX_BEAMS} supported")
        self.temperature, self.beam_size, self.top_k = float(temperature), int(beam_size), int(top_k)
        self.unk_index, self.eos_index, self.device = unk_index, eos_index, device
        self.n_img, self.max_len = n_img, max_len
        self.seed, self.img0, self.noise_source = int(seed), int(img0), noise_source
        if hasattr(noise_source, "attach"):
            noise_source.attach(self)
        # optional device-resident int64 word XOR-ed into the seed by the kernels: lets a captured hipGraph of the
        # whole decode be replayed with a fresh seed (kernel arguments are frozen at capture)
        self.seed_tensor = seed_tensor
        r = n_img * beam_size
        dev = device
        # all zero-initialised state carved out of ONE zeroed arena (one fill launch instead of eight per generate call)
        words = [r * max_len, r, (r + 3) // 4, r, r, (n_img + 3) // 4, n_img, 1]
        offs = [0]
        for w in words:
            offs.append(offs[-1] + (w + 3) // 4 * 4)                                   # 16-byte aligned pieces
        arena = torch.zeros((offs[-1],), dtype=torch.int32, device=dev)
        piece = lambda i: arena[offs[i]:offs[i] + words[i]]
        self.tokens = piece(0).view(r, max_len)
        self.vals = piece(1).view(torch.float32)
        # engine state (uint8, all images).  `has_ended` starts as the SAME tensor; the reference-style method surface below
        # (process_logits, or a caller assigning `helper.has_ended = ...` as rnn_models.py:103,126 do) rebinds only the public name
        self._ended = piece(2).view(torch.uint8)[:r]
        self.has_ended = self._ended
        self._draws = 0                   # multinomial calls made through the method surface (Philox `draw` counter)
        self.parent, self.hparent = piece(3), piece(4)
        self.done = piece(5).view(torch.uint8)[:n_img]
        self.end_step, self.err = piece(6), piece(7)
        self.pick_idx = torch.empty((r, beam_size), dtype=torch.int32, device=dev)
        self.pick_val = torch.empty((r, beam_size), dtype=torch.float32, device=dev)
        # KV-cache ancestor table (Transformer only): src[r, j] = row holding position j of r's history
        self.src = None
        if src_len:
            base = (torch.arange(r, dtype=torch.int32, device=dev) // beam_size) * beam_size
            self.src = base[:, None].expand(r, src_len).contiguous()

    def set_prefix(self, caption):
        """caption int64 [n_img, p]: teacher-forced beginning, copied to every beam row."""
        p = caption.shape[1]
        self.tokens[:, :p] = caption.to(torch.int32).repeat_interleave(self.beam_size, dim=0)

    def _noise(self, kind, step, shape, ld=None):
        if self.noise_source is None:
            return None
        t = self.noise_source(kind, step, shape)
        if t is None:
            return None
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if ld is not None and ld != t.shape[1]:          # row noise is indexed with the logits' (padded) row stride
            buf = torch.ones((t.shape[0], ld), dtype=torch.float32, device=self.device)
            buf[:, :t.shape[1]] = t
            t = buf
        return t

    def step(self, logits, first, write_pos, t, step_index, first_sets_ended=False, group_max=None):
        """One beam step from ``logits`` ([n_img, V] if ``first`` else [n_img*beam, V]):
        beam.py:55-108 + the caller-side candidate draw (rnn_models.py:116-128, transformers.py:557-569)."""
        rows = logits.shape[0]
        rpi = 1 if first else self.beam_size
        assert rows == self.n_img * rpi
        v = logits.shape[1]
        if self.exact:
            group_max = None              # the general sampler reads the whole row
        if group_max is not None and self.top_k <= hip.n_groups(v):   # k group maxima bound the k-th logit
            # bf16 path: the vocabulary GEMM left per-row maxima of every 64-column group (dh_vocab_logits)
            hip.beam_row_sample_groups(logits, v, group_max, rows, rpi, self.beam_size, self.top_k, self.temperature,
                                       self.unk_index, self._noise("row", step_index, (rows, v), logits.stride(0)), self.seed, self.img0,
                                       step_index, self.pick_idx, self.pick_val, self.err, seed_ptr=self.seed_tensor)
        else:
            hip.beam_row_sample(logits, v, rows, rpi, self.beam_size, self.top_k, self.temperature, self.unk_index,
                                self._noise("row", step_index, (rows, v), logits.stride(0)), self.seed, self.img0, step_index,
                                self.pick_idx, self.pick_val, self.err, seed_ptr=self.seed_tensor, exact=self.exact)
        noise = None if first else self._noise("cand", step_index, (self.n_img, self.beam_size ** 2))
        hip.beam_select(self.pick_idx, self.pick_val, self.tokens, self.vals, self._ended, self.src,
                        self.parent, self.hparent, self.done, self.end_step, self.n_img, self.beam_size, first,
                        first_sets_ended, write_pos, t, step_index, self.temperature, self.eos_index, noise,
                        self.seed, self.img0, seed_ptr=self.seed_tensor)

    def finalize(self, len_bias_done, full_len, pad_index=0, defer_check=False, first_beam=False):
        """Final draw among the beams and output copy; returns (tokens int64 [n_img, max_len], lengths).
        ``defer_check``: skip the host read of the device error word (hipGraph capture) -- the caller checks
        ``self.err`` after replay.  ``first_beam``: no draw, beam 0 -- what the reference's final
        ``sample_k_indices(sample_val, k=1)`` degenerates to when ``sample_val`` is still the ``[beam, 1]`` column of the
        first step (rnn_models.py:93, 140-141: no decode step ran because the prefix already fills ``max_len - 1``)."""
        out = torch.empty((self.n_img, self.max_len), dtype=torch.int32, device=self.device)
        out_len = torch.empty((self.n_img,), dtype=torch.int32, device=self.device)
        if first_beam:                 # the kernel's race p / noise with an infinite handicap on every beam but the first
            noise = torch.full((self.n_img, self.beam_size), float("inf"), dtype=torch.float32, device=self.device)
            noise[:, 0] = 1.0
        else:
            noise = self._noise("final", 0, (self.n_img, self.beam_size))
        hip.beam_finalize(self.tokens, self.vals, self.done, self.end_step, out, out_len, self.n_img, self.beam_size,
                          len_bias_done, full_len, pad_index, self.temperature, noise, self.seed, self.img0,
                          seed_ptr=self.seed_tensor)
        if defer_check:
            return out.long(), out_len.long(), self.err
        self.check()
        return out.long(), out_len.long()

    def check(self):
        """Raises like the reference does when every logit of a row was filtered (beam.py:46)."""
        self.raise_for(int(self.err.item()))

    @staticmethod
    def raise_for(code):
        if code == 0:
            return
        if code & hip.ERR_NONFINITE:
            # torch.multinomial's message for a probability row holding NaN (softmax over logits with a NaN or +inf among them)
            raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0 (a row's logits hold NaN or +inf)")
        if code & hip.ERR_ALL_FILTERED:
            raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0 "
                               "(every logit of a row was filtered: <unk> was the only top-k token)")
        if code & hip.ERR_OVERFLOW:
            raise BeamOverflow("more than 1024 logits of a row tie at its top-k threshold (DH_BEAM_MAX_SURVIVORS): repeat with exact=True")
        # hip.ERR_TOO_FEW (fewer positive-probability tokens than beams: top_k == beam_size with <unk> in the top-k, or
        # beam_size >= num_tokens) is NOT an error: torch.multinomial of the torch versions this was pinned against fills the
        # remaining slots with zero-probability tokens, the kernels with <pad> at score -inf -- a dead beam either way, which no
        # later draw can pick.  The randomised sweep (tools/fuzz_generate.py) reproduces the reference token for token in all such
        # cases.  (torch <= 1.x raised "invalid multinomial distribution" here.)  The bit stays readable in ``helper.err``.

    def all_ended(self):
        """Host-visible early-exit test (one sync; callers poll it sparsely, not per token).  Engine use: every image's beams
        have ended (`done`); method-surface use (beam.py:110-112): every flag of the caller-visible ``has_ended``."""
        if self.has_ended is not self._ended:
            return bool(self.has_ended.cpu().numpy().all())
        return bool(self.done.cpu().numpy().all())

    # ---- the reference's method surface (beam.py:32-108): same names, argument meaning, shapes, dtypes and in-place behaviour,
    # each method one HIP launch.  Host-driven like the reference's own loops (rnn_models.py:87-128, transformers.py:532-569):
    # sample_k_indices reads the device error word (one sync per draw) so that it raises where torch.multinomial raises.
    def _draw_noise(self, shape):
        """Exp(1) noise of one multinomial call: the ``noise_source`` hook (``("multinomial", call number, shape)``) or Philox."""
        self._draws += 1
        if self.noise_source is None:
            return None
        t = self.noise_source("multinomial", self._draws - 1, shape)
        return None if t is None else t.to(device=self.device, dtype=torch.float32).reshape(shape).contiguous()

    def filter_top_k(self, logits):
        """beam.py:32-37: IN PLACE -- entries strictly below the row's ``top_k``-th largest value and the ``unk`` column become
        ``-inf`` (ties at the threshold stay); returns its argument."""
        hip.beam_filter_top_k(logits, self.top_k, self.unk_index)
        return logits

    def sample_k_indices(self, logits, k=None):
        """beam.py:39-48: ``torch.multinomial(softmax(logits / temperature), k)`` (no replacement) as an Exp(1) race on the
        device; ``logits`` is ``[n, V]`` -> int64 ``[n, k]`` or 1-D ``[V]`` -> ``[k]`` (the candidate draws, rnn_models.py:120)."""
        k = self.beam_size if k is None else int(k)
        x = logits if logits.dim() == 2 else logits.reshape(1, -1)
        if x.dtype != torch.float32 or x.stride(-1) != 1:
            raise TypeError("sample_k_indices expects fp32 logits with unit column stride")
        out = torch.empty((x.shape[0], k), dtype=torch.int64, device=x.device)
        self.err.zero_()
        hip.beam_sample_k(x, k, self.temperature, self._draw_noise(tuple(x.shape)), self.seed, self.img0, self._draws, out, self.err,
                          seed_ptr=self.seed_tensor)
        code = int(self.err.item())
        self.raise_for(code)            # (fewer positive entries than k: the zero-probability ones follow in index order, as
                                        #  current torch.multinomial returns them in an unspecified order -- no error)
        return out if logits.dim() == 2 else out[0]

    @staticmethod
    def filter_by_indices(values, indices):
        """beam.py:50-53: ``torch.gather(values, 1, indices)``."""
        out = torch.empty(indices.shape, dtype=torch.float32, device=values.device)
        hip.beam_gather(values, indices.contiguous(), out)
        return out

    def process_logits(self, logits, sample_seq, sample_val):
        """beam.py:55-108.  ``logits [n, V]`` fp32 (filtered IN PLACE, as the reference does), ``sample_seq [n, L]`` int64,
        ``sample_val [n]`` or ``[n, 1]``, ``self.has_ended [n]`` -> ``(prev_seqs, prev_vals), (new_ind, new_val)`` over the
        ``n_cand = sum(1 if ended else beam_size)`` candidates; ``self.has_ended`` becomes the ``[n_cand]`` bool flags."""
        b = self.beam_size
        logits = self.filter_top_k(logits)
        new_ind = self.sample_k_indices(logits, k=b)
        gathered = self.filter_by_indices(logits, new_ind)
        ended = self.has_ended
        ended = ended.view(torch.uint8) if ended.dtype == torch.bool else ended.to(torch.uint8)
        ended = ended.to(logits.device).contiguous()
        n = ended.shape[0]
        if logits.shape[0] != n or sample_seq.shape[0] != n:
            raise IndexError(f"process_logits: {logits.shape[0]} logit rows / {sample_seq.shape[0]} sequences for {n} has_ended flags")
        n_cand = int(sum(1 if e else b for e in ended.cpu().tolist()))
        seqs = sample_seq.to(torch.int64).contiguous()
        vals = sample_val.to(torch.float32).contiguous()
        dev = logits.device
        prev_seqs = torch.empty((n_cand, seqs.shape[1]), dtype=torch.int64, device=dev)
        prev_vals = torch.empty((n_cand,) + tuple(vals.shape[1:]), dtype=torch.float32, device=dev)
        out_ind = torch.empty((n_cand,), dtype=torch.int64, device=dev)
        out_val = torch.empty((n_cand,), dtype=torch.float32, device=dev)
        out_ended = torch.empty((n_cand,), dtype=torch.uint8, device=dev)
        hip.beam_expand(new_ind, gathered, ended, seqs, vals, b, self.eos_index, prev_seqs, prev_vals, out_ind, out_val, out_ended)
        self.has_ended = out_ended.view(torch.bool)
        return (prev_seqs, prev_vals), (out_ind, out_val)


def make_noise_source(rng, seed, noise_source, lo, hi, img0, state0=None):
    """The noise source of one decode session (images ``[lo, hi)`` of the batch): the caller's hook, or the torch-generator replay
    of ``rng="torch"`` (``TorchRngNoise``); ``rng`` None / "philox" = the kernels' own counter-based generator."""
    if rng in (None, "philox"):
        return noise_source
    if rng != "torch":
        raise ValueError(f'rng must be None, "philox" or "torch", not {rng!r}')
    if noise_source is not None:
        raise ValueError('rng="torch" and noise_source are mutually exclusive')
    return TorchRngNoise(seed, hi - lo, img0 + lo, state0=state0)


def check_ids(ids, n, capturing_ok=True):
    """``nn.Embedding``'s ``IndexError`` for ids outside ``[0, n)`` (reference: every token / label lookup).  The kernels gather rows by
    these ids without a bounds test, so an id outside the table would read foreign memory (round 5: a GPU memory fault on a label of -3,
    silent garbage on a token of V + 100): checked on the host, two scalars per call.  Inside a hipGraph capture nothing can be read
    back: ``generate_batch_graphed`` checks its inputs before the replay instead."""
    if ids is None or ids.numel() == 0:
        return
    if ids.is_cuda and capturing_ok and torch.cuda.is_current_stream_capturing():
        return
    lo, hi = torch.aminmax(ids)
    if int(lo) < 0 or int(hi) >= n:
        raise IndexError("index out of range in self")


def check_lengths(lengths, steps):
    """``pack_padded_sequence``'s errors for the teacher-forced LSTM forward (reference rnn_models.py:39)."""
    if int(lengths.min()) <= 0:
        raise RuntimeError("Length of all samples has to be greater than 0, but found an element in 'lengths' that is <= 0")
    if int(lengths.max()) > steps:
        raise RuntimeError(f"Expected sequence length to be larger than or equal to the maximum of lengths, but got sequence length {steps} and max length {int(lengths.max())}")


def classifier_must_be_finite(plan):
    """A NaN or inf in the classifier's weight or bias is in every row's logits: the reference's ``torch.multinomial`` raises on the
    first draw (beam.py:46).  The full-row samplers see such a logit (it sorts above everything) and flag ERR_NONFINITE; the
    group-maximum pre-filter of the 16-bit paths would not (``max`` drops NaN), so generate checks the operands once per plan."""
    ok = plan.get("_cls_finite")
    if ok is None:
        ok = plan["_cls_finite"] = bool(torch.isfinite(plan["cls_w"]).all()) and bool(torch.isfinite(plan["cls_b"]).all())
    if not ok:
        raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0 (the classifier's weight or bias holds NaN or inf)")


def call_logits_hook(hook, pos, logits, helper):
    """``logits_hook(pos, logits)`` before the beam step of position ``pos``; a hook whose attribute ``with_tokens`` is true is called
    ``hook(pos, logits, tokens)`` with the engine's int32 ``[rows, max_len]`` token table (row r's own history in columns < pos):
    what a checker needs to re-run a row teacher-forced (tests/test_fullsize_gpu.py)."""
    if getattr(hook, "with_tokens", False):
        hook(pos, logits, helper.tokens)
    else:
        hook(pos, logits)


_overflow_warned = False


def warn_overflow_retry():
    """One warning per process when a batch is decoded a second time through the general sampler (``BeamOverflow``): user hooks
    (``logits_hook``, ``noise_source``) then fire once more for the same steps."""
    global _overflow_warned
    if not _overflow_warned:
        _overflow_warned = True
        import warnings
        warnings.warn("deephumor_amd: a row had more logits tied at its top-k threshold than the pre-filtered samplers hold (flat "
                      "logits); the batch is decoded again with exact=True -- logits_hook / noise_source callbacks run a second time",
                      RuntimeWarning, stacklevel=3)


def resolve_seed(seed, noise_source=None):
    """``seed=None`` (the default of every ``generate``): a fresh 62-bit Philox key drawn from torch's default CPU
    generator -- the generator the reference's ``torch.multinomial`` calls consume (beam.py:46) -- so ``torch.manual_seed``
    controls ``generate`` and successive calls give different captions, as with the reference.  An int is used as is.
    With caller-supplied noise (RNG-replay parity tests) the Philox key is unused and nothing is drawn."""
    if seed is None and noise_source is not None:
        return 0
    if seed is None:
        return int(torch.randint(0, 1 << 62, (), dtype=torch.int64).item())
    return int(seed)


_STREAMS = {}


def run_interleaved(make_session, n_img, n_streams):
    """Runs ``make_session(lo, hi)`` -- a generator that decodes images ``[lo, hi)`` and yields after every
    position, returning ``(tokens, lengths)`` -- either once, or as ``n_streams`` image sub-batches advanced
    round-robin on separate HIP streams.  Decode positions are chains of small, latency-bound kernels
    (a 640-row GEMM fills a fraction of the 256 CUs); two independent chains in flight fill the gaps.
    Captions are unchanged: every image's noise is keyed by its global index (``img0 + lo``)."""
    n_streams = max(1, min(n_streams, n_img))
    if n_streams == 1:
        gen = make_session(0, n_img)
        while True:
            try:
                next(gen)
            except StopIteration as e:
                return e.value
    main = torch.cuda.current_stream()
    dev = main.device
    pool = _STREAMS.setdefault(dev, [])
    while len(pool) < n_streams:
        pool.append(torch.cuda.Stream(device=dev))
    base, extra = divmod(n_img, n_streams)
    bounds, lo = [], 0
    for i in range(n_streams):
        hi = lo + base + (1 if i < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    gens, results, alive = [], [None] * n_streams, set(range(n_streams))
    for (lo, hi), st in zip(bounds, pool):
        st.wait_stream(main)
        gens.append(make_session(lo, hi))
    while alive:
        for i in sorted(alive):
            with torch.cuda.stream(pool[i]):
                try:
                    next(gens[i])
                except StopIteration as e:
                    results[i] = e.value
                    alive.discard(i)
    for st in pool[:n_streams]:
        main.wait_stream(st)
    for r in results:
        for t in r:
            t.record_stream(main)
    out = (torch.cat([r[0] for r in results], 0), torch.cat([r[1] for r in results], 0))
    if len(results[0]) > 2:               # defer_check sessions also return their device error word: OR of the sub-batches' words
        err = results[0][2].clone()
        for r in results[1:]:
            err |= r[2]
        out += (err,)
    return out
