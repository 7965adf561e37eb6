"""Token-by-token decode loop around the quantized model (harness for BASELINE configs 4 and 5).

Counterpart of the reference's ``ChatGLMDecoder.generate`` and ``top_p_sampling``
(chatglm_q/decoder.py:12-27,65-108): batch 1, same sampler, same timing definitions
(``gen = (n - 1) / sum(t[1:])``, ``avg = n / sum(t)``, chatglm_q/decoder.py:99-106).

MI355X-first differences (results unchanged):
  * the key/value cache is preallocated and a decode step is shape-static, so ONE step - 113 QLinear
    launches plus the small ops around them - is captured in a HIP graph and replayed per token; the
    position counter, the cache write index, the attention mask, the choice of the next token - the
    reference's top-k / top-p sampler (``qlinear_top_p_sample``: one launch, counter-based generator in
    device memory) or the argmax - and the next input id are all updated on the device inside that
    graph, so the host only replays it;
  * lm_head is evaluated for the last position only (the loop uses nothing else, decoder.py:85);
  * prefill can be chunked (`prefill_chunk`), each chunk one forward over batch x chunk rows.
Text handling (tokenizer, chat template, punctuation fix-ups) is out of scope: the loop works on token ids;
a tokenizer object with ``encode`` / ``decode`` may be supplied and is used if present.
"""
from __future__ import annotations

import time
from typing import Iterable, Optional

import torch
from torch import Tensor

from .model import ChatGLM2Model, KVCache


def top_p_sampling(logits: Tensor, top_k: int = 100, top_p: float = 0.8, temperature: float = 1.0) -> Tensor:
    """softmax(logits / T) -> keep the top_k most likely -> drop those whose preceding cumulative mass
    exceeds top_p -> renormalise -> one multinomial draw (chatglm_q/decoder.py:12-27)."""
    probs = torch.softmax(logits.float() / temperature, dim=-1)
    probs, indices = torch.sort(probs, dim=-1, descending=True)
    probs, indices = probs[..., :top_k], indices[..., :top_k]
    before = torch.cumsum(probs, dim=-1) - probs
    probs = torch.where(before > top_p, torch.zeros_like(probs), probs)
    probs = probs / probs.sum(dim=-1, keepdim=True)
    pick = torch.multinomial(probs, num_samples=1)
    return torch.gather(indices, dim=-1, index=pick)[..., 0]


def filtered_distribution(logits: Tensor, top_k: int = 100, top_p: float = 0.8, temperature: float = 1.0):
    """The deterministic part of the sampler: (renormalised probabilities, token indices)."""
    probs = torch.softmax(logits.float() / temperature, dim=-1)
    probs, indices = torch.sort(probs, dim=-1, descending=True)
    probs, indices = probs[..., :top_k], indices[..., :top_k]
    before = torch.cumsum(probs, dim=-1) - probs
    probs = torch.where(before > top_p, torch.zeros_like(probs), probs)
    return probs / probs.sum(dim=-1, keepdim=True), indices


# hook of the developer experiments (chatglm_q_amd/dev/experiments.py); None in the product
AHEAD_LAUNCH = True           # greedy graph decoding launches step k + 1 before reading token k (ChatGLMDecoder._generate)
POST_GENERATE_CHECK = None

class DecodeSession:
    """One sequence batch on one device: preallocated cache, eager (chunked) prefill, graph-replayed decode."""

    def __init__(self, model: ChatGLM2Model, batch: int, capacity: int, use_graph: Optional[bool] = None,
                 decode_only: bool = False, low_footprint: bool = False):
        self.model = model
        # low_footprint (round 5): once the derived layouts a phase reads exist, the canonical GPU buffers of every int4 module are
        # dropped (DynamicQuantizeLinear.drop_canonical: state_dict() / save_pretrained rebuild them byte for byte from part 1) -
        # ChatGLM2-6B int4g32: ~8.5 GB resident for prefill + decode, ~3.5 GB with decode_only (the reference's "6G+", readme.md:72)
        self.low_footprint = low_footprint
        # decode_only: before the decode step is captured, free the derived layouts only prefill / batched rows use (part 2 of
        # every int4 layer, the un-gated part 1 of w_in once its gate-interleaved copy exists): ~10.8 -> ~6.6 GB resident for
        # ChatGLM2-6B int4g32.  A later prefill rebuilds what it needs (and the graph is re-captured).
        self.decode_only = decode_only
        p = model.final_ln.weight
        self.device, self.dtype = p.device, p.dtype
        if capacity > model.config.max_sequence_length:
            raise ValueError(f"cache capacity {capacity} exceeds max_sequence_length {model.config.max_sequence_length} "
                             "(no rotary table rows for those positions)")
        self.batch, self.capacity = batch, capacity
        self.cache: KVCache = model.new_cache(batch, capacity)
        self.use_graph = (self.device.type == "cuda") if use_graph is None else use_graph
        dev = self.device
        # device-resident step state (static addresses: the captured graph reads and updates them)
        self.tok = torch.zeros(batch, 1, dtype=torch.long, device=dev)
        self.write_index = torch.zeros(1, dtype=torch.long, device=dev)
        self.pos = torch.zeros(batch, 1, dtype=torch.long, device=dev)
        self.mask = torch.full((batch, 1, capacity), -1e10, dtype=torch.float32, device=dev)
        self.pad_cols = torch.zeros(batch, capacity, dtype=torch.bool, device=dev)   # left-padding columns (never attended)
        self.n_tokens = torch.zeros(batch, dtype=torch.long, device=dev)             # real tokens per sequence so far
        self.logits: Optional[Tensor] = None
        self.graph = None                                       # the captured step of the mode last asked for ...
        self._graphs = {}                                       # ... of every mode captured under the current layouts
        self._captured_mode = None
        self._graph_key = None
        # the device sampler's state (chatglm_q/decoder.py:12-27 inside the captured step): top_k / top_p / temperature and the
        # generator's (seed, per-row draw counter) live in device memory, so one captured graph serves every setting and every seed
        self.sample_params = torch.tensor([100.0, 0.8, 1.0], dtype=torch.float32, device=dev)
        self.rng_state = torch.zeros(1 + batch, dtype=torch.long, device=dev)
        self.length = 0
        self.busy = False                                       # a generate_ids generator is running on this session

    def reset(self):
        """Forget the sequences (not the captured graph): the session can serve the next generation.  Cache rows need no
        clearing - every position is masked until a prefill / decode step writes and unmasks it."""
        self.length = 0
        self.cache.length = 0
        self.pad_cols.zero_()
        self.n_tokens.zero_()
        self.write_index.zero_()
        self.pos.zero_()
        self.mask.fill_(-1e10)
        return self

    # -- prefill -------------------------------------------------------------------------------------
    @torch.no_grad()
    def prefill(self, ids: Tensor, chunk: Optional[int] = None, attention_mask: Optional[Tensor] = None) -> Tensor:
        """ids (batch, S).  Returns the last position's logits (batch, vocab).

        ``attention_mask`` (batch, S), 1 = token / 0 = pad, follows the reference's convention for batches of unequal
        length (chatglm_q/model.py:297-318): sequences are LEFT padded, a pad column is masked for every query
        (``causal | ~attention_mask``) and a token's rotary position is its count among the real tokens
        (``cumsum(attention_mask)``: 1-based, pads get 0).  The pad columns stay masked through later chunks and
        decode steps."""
        B, S = ids.shape
        if S + self.length > self.capacity:
            raise ValueError(f"sequence {S + self.length} exceeds the cache capacity {self.capacity}")
        ids = ids.to(self.device)
        if attention_mask is None:
            valid = torch.ones(B, S, dtype=torch.bool, device=self.device)
        else:
            if attention_mask.shape != ids.shape:
                raise ValueError(f"attention_mask {tuple(attention_mask.shape)} != ids {tuple(ids.shape)}")
            valid = attention_mask.to(self.device).bool()
        self.pad_cols[:, self.length:self.length + S] = ~valid
        positions = self.n_tokens[:, None] + torch.cumsum(valid.long(), dim=1)       # 1-based among real tokens
        positions = torch.where(valid, positions, torch.zeros_like(positions))       # cumsum of a left pad is 0
        # default: the whole prompt in one pass while that is at most 8192 rows per QLinear call (where the prefill GEMMs are at their
        # best and the (batch, chunk, keys) mask stays small), otherwise chunks of 8192 // batch positions
        chunk = chunk or (S if B * S <= 8192 else max(256, 8192 // B))
        t = torch.arange(self.capacity, device=self.device)
        logits = None
        for s0 in range(0, S, chunk):
            s1 = min(S, s0 + chunk)
            rows = torch.arange(self.length + s0, self.length + s1, device=self.device)
            blocked = (t[None, None, :] > rows[None, :, None]) | self.pad_cols[:, None, :]
            mask = blocked.float() * -1e10
            logits = self.model.step(ids[:, s0:s1], self.cache, rows, positions[:, s0:s1], mask,
                                     last_only=True, kv_len=self.length + s1)
        self.length += S
        self.cache.length = self.length
        self.n_tokens += valid.long().sum(dim=1)
        # arm the decode-step state
        self.write_index.fill_(self.length)
        self.pos.copy_((self.n_tokens + 1)[:, None])            # positions are 1-based (model.py:308)
        self.mask.fill_(-1e10)
        self.mask[:, :, : self.length + 1] = 0.0                # incl. the position the next step writes
        self.mask.masked_fill_(self.pad_cols[:, None, :], -1e10)
        if self.low_footprint:
            self.drop_canonical()                               # the layouts this prefill read exist now: the canonical copies can go
        return logits[:, -1]

    # -- one decode step -------------------------------------------------------------------------------
    def set_sampling(self, top_k: int = 100, top_p: float = 0.8, temperature: float = 1.0, seed: Optional[int] = None):
        """Arm the device sampler: its parameters and (seed given) a fresh generator state - draw counters back to 0."""
        if top_k < 1 or not temperature > 0:
            raise ValueError("top_k >= 1 and temperature > 0")
        self.sample_params.copy_(torch.tensor([float(top_k), float(top_p), float(temperature)], dtype=torch.float32))
        if seed is not None:
            st = torch.zeros(1 + self.batch, dtype=torch.long)
            st[0] = int(seed) & 0x7FFFFFFFFFFFFFFF
            self.rng_state.copy_(st)

    def device_sampler_serves(self, top_k: int) -> bool:
        """The one-launch sampler runs on a GPU for top_k <= 1024 (or a vocabulary that small); otherwise the composed torch ops."""
        from . import fused_ops
        return self.device.type == "cuda" and (top_k <= fused_ops.SAMPLER_MAX_TOP_K
                                               or self.model.config.vocab_size <= fused_ops.SAMPLER_MAX_TOP_K)

    def sample_first(self, logits: Tensor) -> Tensor:
        """The token after a prefill, drawn by the device sampler from the prefill's last-position logits (no bookkeeping: prefill
        armed the step state).  Returns the (batch,) device tensor; ``tok`` holds it as the next step's input."""
        from . import fused_ops
        lg = logits if logits.stride(-1) == 1 else logits.contiguous()
        fused_ops.top_p_sample(lg, self.tok.view(-1), self.rng_state, dev_params=self.sample_params)
        return self.tok.view(-1)

    def _step_body(self, greedy: bool, sample: bool = False):
        # invariant on entry: mask is 0 for positions <= write_index (the new position may attend to itself)
        logits = self.model.step(self.tok, self.cache, self.write_index, self.pos, self.mask, last_only=True)
        self.logits = logits[:, -1]
        if sample:                                              # the reference's sampler + the bookkeeping: one launch
            from . import fused_ops
            lg = self.logits if self.logits.stride(-1) == 1 else self.logits.contiguous()
            fused_ops.top_p_sample(lg, self.tok.view(-1), self.rng_state, dev_params=self.sample_params,
                                   write_index=self.write_index, pos=self.pos.view(-1), mask=self.mask)
            return
        if greedy and self.logits.is_cuda and self.logits.stride(-1) == 1:
            from . import fused_ops
            fused_ops.greedy_advance(self.logits, self.tok, self.write_index, self.pos, self.mask)   # one launch
            return
        if greedy:
            self.tok.copy_(self.logits.argmax(dim=-1, keepdim=True))
        self.write_index.add_(1)
        self.pos.add_(1)
        self.mask.index_fill_(2, torch.clamp(self.write_index, max=self.capacity - 1), 0.0)

    def drop_canonical(self) -> int:
        """Low-footprint mode: free the canonical buffers of every int4 module that can serve without them, and the plain copies of a
        first MLP projection that its gate-interleaved copies have made redundant (the fused step and the gated prefill GEMM read the
        gated part 1 / part 2; anything that asks for the plain ones again rebuilds them from what stays).  Returns the bytes freed."""
        freed = 0
        for m in self.model.modules():
            if not hasattr(m, "drop_canonical"):
                continue
            freed += m.drop_canonical()
            nb = m.derived_nbytes()
            parts = [plain for plain, gated in (("packed", "gated"), ("tiled", "gated_tiled")) if nb.get(plain) and nb.get(gated)]
            if parts and m.canonical_dropped:
                freed += sum(nb[k] for k in parts)
                m.release(*parts)
        return freed

    def release_prefill_layouts(self) -> int:
        """Free the derived layouts a one-row decode step never reads; returns the bytes released."""
        freed = 0
        for m in self.model.modules():
            if not (hasattr(m, "release") and hasattr(m, "derived_nbytes")):
                continue
            nb = m.derived_nbytes()
            parts = [k for k in ("tiled", "gated_tiled") if nb.get(k)]
            if nb.get("gated") and nb.get("packed") and "gated_tiled" in nb:      # int4 first MLP projection: the fused step reads the gated copy
                parts.append("packed")
            if parts:
                freed += sum(nb[k] for k in parts)
                m.release(*parts)
        return freed

    def _layout_fingerprint(self):
        """What a captured step bakes in besides this session's own buffers: the canonical buffers' identity + version and the
        derived layouts' addresses of every quantized module, and which path (``act_quant``) each one takes.  A graph captured
        under another fingerprint would replay against freed / stale derived buffers (``load_state_dict`` /
        ``apply_weights_`` / ``invalidate()`` drop them, the next forward allocates new ones) - ADVICE r2."""
        from . import _lib
        key = []
        for m in self.model.modules():
            if hasattr(m, "weight_scale") and hasattr(m, "weight"):
                derived = tuple(None if t is None else (t[0] if isinstance(t, tuple) else t).data_ptr()
                                for t in (getattr(m, a, None) for a in ("_packed", "_tiled", "_gated", "_gated_tiled", "_a8")))
                key.append((_lib.buffer_key(m.weight, m.weight_scale, getattr(m, "bias", None)), derived,
                            getattr(m, "act_quant", None)))
        return tuple(key)

    @staticmethod
    def _mode(greedy: bool, sample: bool) -> str:
        return "sample" if sample else ("greedy" if greedy else "logits")

    @property
    def _captured_greedy(self):                                 # kept for the developer tools that read it
        return self._captured_mode == "greedy"

    @torch.no_grad()
    def capture(self, greedy: bool = True, sample: bool = False):
        """Capture one decode step of the mode - "sample": the device sampler picks the next token, "greedy": the argmax does,
        "logits": the host decides - or keep the captured one while nothing it bakes in has changed.  Graphs of several modes live
        side by side (a decoder that alternates greedy and sampled generations captures each once)."""
        if not self.use_graph:
            return
        from . import _lib
        mode = self._mode(greedy, sample)
        if self._graphs:
            # cheap test first (a process-wide counter every derived-layout build / drop bumps), the full fingerprint when it moved
            valid = self._graph_epoch == _lib.layout_epoch() and self._graph_versions == self._canonical_versions()
            if not valid and self._graph_key == self._layout_fingerprint():
                self._graph_epoch, self._graph_versions = _lib.layout_epoch(), self._canonical_versions()
                valid = True
            if not valid:
                self._graphs.clear()                            # weights / layouts moved under the graphs: re-capture
                self.graph = self._captured_mode = None
            elif mode in self._graphs:
                self.graph, self._captured_mode = self._graphs[mode], mode   # a reused session keeps its graphs (static addresses)
                return
        if self.decode_only:
            self.release_prefill_layouts()
        saved = (self.tok.clone(), self.write_index.clone(), self.pos.clone(), self.mask.clone(), self.rng_state.clone())
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._step_body(greedy, sample)                     # warm-up: allocations, lazy repacks
            # the warm-up built what the step reads (e.g. the gate-interleaved copy of w_in): what it made redundant goes now,
            # and the step runs once more so that every pre-bound launch exists before the capture
            if self.decode_only and self.release_prefill_layouts():
                self._step_body(greedy, sample)
            if self.low_footprint and self.drop_canonical():
                self._step_body(greedy, sample)                 # the pre-bound launches re-validate against the stand-in buffers
        torch.cuda.current_stream(self.device).wait_stream(side)
        if self._graphs and self._graph_epoch != _lib.layout_epoch():
            self._graphs.clear()                                # this mode's warm-up rebuilt / dropped a layout the older graphs baked in
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._step_body(greedy, sample)
        # capture ran the body twice on live state: restore (the cache rows it touched get rewritten)
        for dst, src in zip((self.tok, self.write_index, self.pos, self.mask, self.rng_state), saved):
            dst.copy_(src)
        self._graphs[mode] = graph
        self.graph, self._captured_mode = graph, mode
        self._graph_key = self._layout_fingerprint()            # AFTER the warm-up built the lazy layouts
        self._graph_epoch, self._graph_versions = _lib.layout_epoch(), self._canonical_versions()

    def _canonical_versions(self):
        """Sum of the canonical buffers' version counters: moves on every in-place write autograd's bookkeeping sees
        (``copy_`` from a loader) even when no derived layout has been rebuilt yet."""
        total = 0
        for m in self.model.modules():
            if hasattr(m, "weight_scale") and hasattr(m, "weight"):
                for t in (m.weight, m.weight_scale, getattr(m, "bias", None)):
                    if t is not None:
                        try:
                            total += t._version
                        except RuntimeError:
                            pass
        return total

    @torch.no_grad()
    def decode_step(self, token: Optional[Tensor] = None, greedy: bool = True, sample: bool = False) -> Tensor:
        """Advance by one token.  `token` (batch, 1) overrides the device-resident next id (needed when the
        host samples).  ``sample``: the device sampler (set_sampling) draws the next id into ``tok``.  Returns this step's
        last-position logits (batch, vocab)."""
        if self.length + 1 > self.capacity:
            raise ValueError("cache capacity exhausted")
        if token is not None:
            self.tok.copy_(token.to(self.device))
        mode = self._mode(greedy, sample)
        if mode in self._graphs:
            from . import _lib
            if self._graph_epoch != _lib.layout_epoch() or self._captured_mode != mode:   # a derived layout was rebuilt / dropped since
                self.capture(greedy, sample)                    # the capture: capture() compares the fingerprint and keeps the graphs
            self.graph.replay()                                 # when only modules of OTHER models / sessions moved
        else:
            self._step_body(greedy, sample)
        self.length += 1
        self.cache.length = self.length
        return self.logits


class SentencePieceIds:
    """Token ids the way the reference's tokenizer produces them (chatglm_q/tokenizer.py:24-64), nothing else: the sentencepiece
    vocabulary followed by the five special tokens, ``encode`` = "[gMASK]" "<sop>" + pieces, ``decode`` drops every id past the
    sentencepiece vocabulary, ``tok[name]`` resolves special tokens and pieces.  The chat template and the punctuation fix-ups of
    the reference's text pipeline are out of scope."""
    SPECIAL = ("[MASK]", "[gMASK]", "[sMASK]", "<sop>", "<eop>")

    def __init__(self, model_file):
        import sentencepiece
        self.model_file = str(model_file)
        self.sp = sentencepiece.SentencePieceProcessor(model_file=self.model_file)
        self.n_pieces = len(self.sp)

    def __len__(self):
        return self.n_pieces + len(self.SPECIAL)

    def __getitem__(self, token: str) -> int:
        if token in self.SPECIAL:
            return self.n_pieces + self.SPECIAL.index(token)
        pid = self.sp.piece_to_id(token)
        if pid == self.sp.unk_id() and self.sp.id_to_piece(pid) != token:
            raise KeyError(token)
        return pid

    def encode(self, text: str, add_special_tokens: bool = True):
        ids = list(self.sp.encode(text))
        return [self["[gMASK]"], self["<sop>"]] + ids if add_special_tokens else ids

    def decode(self, ids) -> str:
        return self.sp.decode([int(i) for i in ids if int(i) < self.n_pieces])


def tok_file_of(tokenizer):
    """The sentencepiece file a tokenizer object was built from, when it says so (this module's wrapper: ``model_file``; the
    reference's ChatGLM2Tokenizer: ``vocab_file``)."""
    for attr in ("model_file", "vocab_file"):
        f = getattr(tokenizer, attr, None)
        if f is not None:
            return f
    return None


def _existing(p):
    return p if p.exists() else None


class ChatGLMDecoder:
    def __init__(self, config, model: ChatGLM2Model, tokenizer=None, eos_token_id: Optional[int] = None, device=None,
                 max_sequence_length: Optional[int] = None, time_log: bool = False, low_footprint: bool = False):
        # low_footprint: the decoder's sessions drop the canonical GPU buffers of every int4 module once the derived layouts a phase
        # reads exist (DecodeSession(low_footprint=True): state_dict() / save_pretrained rebuild them byte for byte)
        self.low_footprint = low_footprint
        self.config = config
        self.model = model
        self.tokenizer = tokenizer
        self.device = device
        self.eos_token_id = eos_token_id
        self.max_sequence_length = max_sequence_length or model.config.max_sequence_length
        self.time_log = time_log
        self.last_stats: dict = {}

    @staticmethod
    def from_pretrained(path, device=None, torch_dtype=None, tokenizer=None, eos_token: str = "</s>", time_log: bool = False,
                        low_footprint: Optional[bool] = None):
        """A reference-format checkpoint folder (``config.json`` + safetensors shards, chatglm_q/loader.py:69-110) -> a decoder
        whose ``generate`` / ``generate_ids`` run this build's fused path by default: preallocated cache, 5 launches per layer,
        one HIP graph per token on a GPU (chatglm_q/decoder.py:49-58 is the reference's constructor of the same name; the hub
        download it falls back to needs a network and is not offered).  ``device`` defaults to the GPU when there is one.
        ``tokenizer``: any object with ``encode`` / ``decode``; when omitted and the folder's sentencepiece file can be read, a
        plain SentencePiece wrapper is used (the reference's chat markers and punctuation fix-ups are text handling, out of scope).
        ``low_footprint`` (None = yes on a GPU): a decoder loaded for generation keeps ONE copy of every int4 weight per consumer -
        the canonical buffers are dropped once the derived layouts exist and rebuilt byte for byte when ``save_pretrained`` /
        ``state_dict()`` ask for them: ChatGLM2-6B int4g32 resident 7.1 GB (prefill + decode layouts) instead of 11.85 GB; the reference's
        guidance is "6G+" (readme.md:72); ``DecodeSession(decode_only=True, low_footprint=True)`` goes to 3.4 GB for decode-only serving."""
        from pathlib import Path
        from .loader import load_model
        path = Path(path)
        if not path.is_dir():
            raise FileNotFoundError(f"{path}: not a checkpoint folder (hub ids are not resolved: no network access in this build)")
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        config, model = load_model(path, torch_dtype, device)
        model.eval()
        eos_id = None
        tokenizer_path = None
        if tokenizer is None:
            tok_file = path / config.tokenizer_file
            if tok_file.exists():
                try:
                    tokenizer = SentencePieceIds(tok_file)
                except (ImportError, OSError, RuntimeError) as e:   # no sentencepiece module / unreadable or corrupt model file
                    # (sentencepiece raises RuntimeError on a file it cannot parse): the decoder still serves token ids
                    import warnings
                    warnings.warn(f"{tok_file}: no tokenizer built ({e}); generate() takes and yields token ids only")
                else:
                    tokenizer_path = tok_file
                    try:
                        eos_id = tokenizer[eos_token]
                    except KeyError:
                        eos_id = None
        elif hasattr(tokenizer, "__getitem__"):
            try:
                eos_id = tokenizer[eos_token]               # the reference tokenizer's lookup (chatglm_q/decoder.py:44)
            except (KeyError, TypeError):
                eos_id = None
        if low_footprint is None:
            low_footprint = torch.device(device).type == "cuda"
        dec = ChatGLMDecoder(config, model, tokenizer, eos_token_id=eos_id, device=device,
                             max_sequence_length=config.model_config.max_sequence_length, time_log=time_log, low_footprint=low_footprint)
        dec.tokenizer_file = tokenizer_path if tokenizer_path is not None else (tok_file_of(tokenizer) or _existing(path / config.tokenizer_file))
        return dec

    def save_pretrained(self, path, shard: bool = True, tokenizer_file=None):
        """chatglm_q/decoder.py:60-61: the folder ``from_pretrained`` (here or in the reference) reads back - ``config.json``, the
        safetensors shards and the sentencepiece file (``tokenizer_file``, default: the one ``from_pretrained`` loaded from; the
        reference's loader opens it unconditionally, chatglm_q/loader.py:85-88).  Without one the folder holds the model only and
        a warning says so."""
        from .loader import ChatGLMLoadConfig, save_model
        if not isinstance(self.config, ChatGLMLoadConfig):
            raise TypeError("save_pretrained needs the ChatGLMLoadConfig the decoder was built with")
        tok = tokenizer_file if tokenizer_file is not None else getattr(self, "tokenizer_file", None)
        if tok is None:
            import warnings
            warnings.warn("save_pretrained: no tokenizer file known (pass tokenizer_file=): the folder will hold the model only and the "
                          "reference's load_model_and_tokenizer will not open it")
        save_model(path, self.config, self.model, shard=shard, tokenizer_file=tok)

    def _session_for(self, capacity: int, use_graph: Optional[bool]) -> DecodeSession:
        """One DecodeSession (cache + captured HIP graph) is kept and reused while the requested capacity fits and the
        model's parameters have not moved; a generation then costs no allocation and no re-capture."""
        p = self.model.final_ln.weight
        key = (p.device, p.dtype, p.data_ptr(), use_graph)
        sess = getattr(self, "_session", None)
        if sess is not None and self._session_key == key and sess.capacity >= capacity and not sess.busy:
            return sess.reset()
        if sess is not None and sess.busy:                      # an unfinished generator still owns it: leave it alone
            return DecodeSession(self.model, 1, capacity, use_graph, low_footprint=self.low_footprint)
        self._session, self._session_key = DecodeSession(self.model, 1, capacity, use_graph, low_footprint=self.low_footprint), key
        return self._session

    @torch.no_grad()
    def generate_ids(self, prefix_ids: Iterable[int], max_generated_tokens: int = 400, top_k: int = 100,
                     top_p: float = 0.8, temperature: float = 1.0, greedy: bool = False, ignore_eos: bool = False,
                     prefill_chunk: Optional[int] = None, use_graph: Optional[bool] = None, sync_every_token: bool = True,
                     seed: Optional[int] = None, device_sampler: Optional[bool] = None):
        """Yields generated token ids one by one (batch 1, like the reference, chatglm_q/decoder.py:70).

        Default = the reference's only mode: top-k / top-p sampling (top_k 100, top_p 0.8, temperature 1.0).  On a GPU the
        sampler is one launch inside the captured step (``device_sampler`` None / True; False: the composed torch ops of
        ``top_p_sampling`` on the host's side of the loop, as the reference runs them) and draws from a counter-based generator
        seeded by ``seed`` (None: a seed taken from torch's global generator, so ``torch.manual_seed`` makes a run repeatable as it
        does for the reference's ``torch.multinomial``).  ``greedy=True``: argmax instead."""
        prefix = list(prefix_ids)
        limit = min(self.max_sequence_length, self.model.config.max_sequence_length)   # the cache cannot outgrow the rotary table
        budget = min(max_generated_tokens, limit - len(prefix))
        if budget <= 0:
            return
        capacity = min(-(-(len(prefix) + budget) // 64) * 64, self.model.config.max_sequence_length)
        sess = self._session_for(capacity, use_graph)
        sess.busy = True
        try:
            dev_sample = (not greedy) and sess.device_sampler_serves(top_k) and device_sampler is not False
            if device_sampler and not greedy and not dev_sample:
                raise ValueError(f"device_sampler=True needs a GPU and top_k <= 1024 (got {sess.device}, top_k {top_k})")
            if dev_sample:
                if seed is None:
                    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                sess.set_sampling(top_k, top_p, temperature, seed)
            yield from self._generate(sess, prefix, budget, top_k, top_p, temperature, greedy, ignore_eos, prefill_chunk,
                                      sync_every_token, dev_sample)
        finally:
            sess.busy = False
            if POST_GENERATE_CHECK is not None:                 # set by chatglm_q_amd.dev.experiments: error words of one-launch MLPs
                POST_GENERATE_CHECK(sess)

    def _generate(self, sess: DecodeSession, prefix, budget: int, top_k: int, top_p: float, temperature: float, greedy: bool,
                  ignore_eos: bool, prefill_chunk: Optional[int], sync_every_token: bool, dev_sample: bool = False):
        # `auto`: the device picks the next token inside the step (argmax or the one-launch sampler) and feeds it back itself
        auto = greedy or dev_sample
        times = []
        sync = (lambda: torch.cuda.synchronize(sess.device)) if sess.device.type == "cuda" else (lambda: None)

        t0 = time.perf_counter()
        logits = sess.prefill(torch.tensor([prefix], dtype=torch.long), prefill_chunk)
        if dev_sample:
            nxt = sess.sample_first(logits)
        else:
            nxt = logits.argmax(-1) if greedy else top_p_sampling(logits, top_k, top_p, temperature)
        token = int(nxt.item())                                    # the sync that makes the timing valid
        times.append(time.perf_counter() - t0)
        generated = [token]
        yield token

        device_loop = auto and not sync_every_token and (ignore_eos or self.eos_token_id is None)
        if len(generated) < budget and (ignore_eos or token != self.eos_token_id):
            sess.tok.fill_(token)
            sess.capture(greedy=greedy, sample=dev_sample)
        pending = []
        # Greedy or device-sampled decoding on a captured step: the graph feeds its own choice back, so step k + 1 does not wait for the HOST to have seen
        # token k.  It is launched before token k is read (one step of speculation: wasted - and harmless, the next prefill re-arms the
        # session - when token k turns out to be the end token), so the device never idles for the host's read-back + launch latency
        # (~20 us of a 1.15 ms token).  Tokens are yielded one by one as before.
        ahead = (auto and not device_loop and sess.graph is not None and sess.device.type == "cuda" and AHEAD_LAUNCH
                 and len(generated) < budget and (ignore_eos or token != self.eos_token_id))
        if ahead:
            ring = torch.empty((2, 1), dtype=torch.long).pin_memory()
            events = [torch.cuda.Event(), torch.cuda.Event()]

            def launch(k):
                sess.decode_step(None, greedy=greedy, sample=dev_sample)
                ring[k & 1].copy_(sess.tok.view(-1), non_blocking=True)
                events[k & 1].record()

            t0 = time.perf_counter()
            k = 0
            launch(0)
            while True:
                more = len(generated) + 1 < budget and sess.length + 1 <= sess.capacity     # a token after this one is wanted (and fits)
                if more:
                    launch(k + 1)
                events[k & 1].synchronize()
                token = int(ring[k & 1, 0])
                now = time.perf_counter()
                times.append(now - t0)
                t0 = now
                generated.append(token)
                yield token
                if not more or (not ignore_eos and token == self.eos_token_id):
                    break
                k += 1
        while not ahead and len(generated) < budget and (ignore_eos or generated[-1] != self.eos_token_id):
            t0 = time.perf_counter()
            if device_loop:
                # the graph feeds its own choice back as the next input: the host only replays
                sess.decode_step(greedy=greedy, sample=dev_sample)
                pending.append(sess.tok.clone())
                generated.append(-1)
                times.append(time.perf_counter() - t0)
                continue
            logits = sess.decode_step(None if auto else torch.tensor([[generated[-1]]]), greedy=greedy, sample=dev_sample)
            nxt = sess.tok[:, 0] if auto else top_p_sampling(logits, top_k, top_p, temperature)
            token = int(nxt.item())
            times.append(time.perf_counter() - t0)
            generated.append(token)
            yield token
        if device_loop and pending:
            t0 = time.perf_counter()
            sync()
            times[-1] += time.perf_counter() - t0
            toks = [int(t.item()) for t in pending]
            generated[-len(toks):] = toks
            for t in toks:
                yield t

        n = len(times)
        rest = sum(times[1:])
        self.last_stats = {
            "prefix": len(prefix), "generated": n, "init_s": times[0], "sum_s": sum(times),
            "gen_tok_per_s": (n - 1) / rest if n > 1 and rest > 0 else float("nan"),
            "avg_tok_per_s": n / sum(times),
        }
        if self.time_log:
            s = self.last_stats
            print("Decoder:")
            print(f"  len: {s['prefix']}(prefix) + {s['generated']}(gen)")
            print(f" init: {s['init_s']:.6f} s")
            print(f"  sum: {s['sum_s']:.6f} s")
            print(f"  gen: {s['gen_tok_per_s']:.6f} tok/s")
            print(f"  avg: {s['avg_tok_per_s']:.6f} tok/s")

    def generate(self, prefix_text: str, **kw):
        """Text interface when a tokenizer (``encode`` / ``decode``) was supplied; yields the decoded text so far."""
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: use generate_ids() with token ids")
        ids = []
        for tok in self.generate_ids(self.tokenizer.encode(prefix_text), **kw):
            ids.append(tok)
            yield self.tokenizer.decode(ids)
