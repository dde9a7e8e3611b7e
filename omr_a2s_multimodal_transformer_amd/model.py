"""Transformer / MultimodalTransformer on the HIP path -- the reference's src/transformer/model.py class surface
(constructor signatures, attributes, hooks, state-dict names) with extra keyword `config` (ModelConfig)
for the parameterised builds BASELINE.json names.  The three reference quirks are reproduced.
"""
from __future__ import annotations

import ctypes
import math
import random
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import functional as Fn
from . import kernels as K
from ._lib import lib
from .config import ModelConfig
from .decoder import Decoder, MultiheadAttention
from .encoder import HEIGHT_REDUCTION, WIDTH_REDUCTION, Encoder
from .lightning_shim import LightningModule
from .metrics import compute_metrics, compute_metrics_sharded
from .runtime import FlatModuleMixin
from .synthetic import EOS_TOKEN, SOS_TOKEN

NUM_CHANNELS = 1  # preprocessing.py:12

_DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16}


def sinusoid_2d(num_channels: int, max_height: int, max_width: int) -> torch.Tensor:
    """model.py:33-42 -> [1, C, max_height, max_width]."""
    half = num_channels // 2
    pos_h = torch.arange(max_height).unsqueeze(1)
    pos_w = torch.arange(max_width).unsqueeze(1)
    den = torch.pow(10000, torch.arange(0, half, 2) / num_channels)
    pe = torch.zeros(1, max_height, max_width, num_channels)
    pe[0, :, :, 0:half:2] = torch.sin(pos_w / den).unsqueeze(0).repeat(max_height, 1, 1)
    pe[0, :, :, 1:half:2] = torch.cos(pos_w / den).unsqueeze(0).repeat(max_height, 1, 1)
    pe[0, :, :, half::2] = torch.sin(pos_h / den).unsqueeze(1).repeat(1, max_width, 1)
    pe[0, :, :, half + 1::2] = torch.cos(pos_h / den).unsqueeze(1).repeat(1, max_width, 1)
    return pe.permute(0, 3, 1, 2).contiguous()


class PositionalEncoding2D(nn.Module):
    """model.py:18-48.  forward takes the encoder's [B,C,h,w] output (channels_last-strided: no copy)."""

    def __init__(self, num_channels: int, max_height: int, max_width: int, dropout_p: float = 0.1) -> None:
        super().__init__()
        self.dropout_p = dropout_p
        pe = sinusoid_2d(num_channels, max_height, max_width)
        self.register_buffer("pe", pe)
        self.register_buffer("pe_hwc", pe[0].permute(1, 2, 0).contiguous(), persistent=False)  # kernel layout [maxh][maxw][C]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xn = x.permute(0, 2, 3, 1)
        if not xn.is_contiguous():
            xn = xn.contiguous()
        y = Fn.AddPE2DFn.apply(xn, self.pe_hwc)
        if self.training and self.dropout_p > 0:
            from .runtime import next_seed
            y = Fn.DropoutFn.apply(y, self.dropout_p, next_seed("nhwc", self.dropout_p), False, False)
        return y.permute(0, 3, 1, 2)


def _h2d(t: torch.Tensor, device) -> torch.Tensor:
    """Host -> device without blocking the host: a blocking copy would wait for the whole previous step (it drains the
    stream) and stop the host from issuing the next step's launches while the GPU is still busy."""
    if t.device == device:
        return t
    if not t.is_cuda and not t.is_pinned() and torch.cuda.is_available():
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


def _flatten_memory(x: torch.Tensor) -> torch.Tensor:
    """x.flatten(2).permute(0, 2, 1).contiguous() (model.py:147): free for channels_last-strided maps."""
    return x.flatten(2).permute(0, 2, 1).contiguous()


class CrossEntropyLoss(nn.Module):
    """CrossEntropyLoss(ignore_index) on [B, V, T] logits (model.py:109,166) through the HIP CE kernels."""

    def __init__(self, ignore_index: int = 0):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits_bvt: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return Fn.cross_entropy(logits_bvt, target, self.ignore_index)


class _Base(FlatModuleMixin, LightningModule):
    def _common_init(self, w2i, i2w, ytest_i2w, max_seq_len, attn_window, teacher_forcing_prob, config: Optional[ModelConfig]):
        if isinstance(config, dict):
            config = ModelConfig.from_dict(config)
        self.config = config if config is not None else ModelConfig()
        if "config" in self.hparams:
            self.hparams["config"] = self.config.to_dict()   # plain data: checkpoints load with weights_only=True
        self._default_compute_dtype = _DTYPES[self.config.compute_dtype]
        self.w2i, self.i2w = w2i, i2w
        self.ytest_i2w = ytest_i2w if ytest_i2w is not None else i2w
        self.padding_idx = w2i["<PAD>"]
        self.max_seq_len = max_seq_len
        self.teacher_forcing_prob = teacher_forcing_prob
        self.attn_window = attn_window
        self.compute_loss = CrossEntropyLoss(ignore_index=self.padding_idx)
        self.Y: List[List[str]] = []
        self.YHat: List[List[str]] = []

    def _make_decoder(self) -> Decoder:
        c = self.config
        dec = Decoder(output_size=len(self.w2i), max_seq_len=self.max_seq_len, num_embeddings=len(self.w2i), embedding_dim=c.d_model,
                      padding_idx=self.padding_idx, ff_dim=c.ff_dim, dropout_p=c.dropout, nhead=c.nhead,
                      num_transformer_layers=c.num_layers, attn_window=self.attn_window)
        dec.fp8_weights = bool(c.fp8_decode)         # KV-cached decoding on the fp8 MFMA (decoder.DecodeState)
        return dec

    # ---- data parallel (ddp.py): bucket boundaries are autograd nodes placed where a bucket's gradients are final
    _reducer = None

    def attach_reducer(self, process_group=None):
        """One bucket per top-level sub-module (encoder(s) | decoder+mixer), reduced as backward leaves it."""
        from .ddp import GradReducer
        flat = self.ensure_flat()
        enc_names = [n for n in flat.names if n.startswith(("encoder.", "image_encoder.", "audio_encoder."))]
        dec_names = [n for n in flat.names if n not in set(enc_names)]
        buckets = [flat.slice_of(enc_names), flat.slice_of(dec_names)]
        self._reducer = GradReducer(flat, process_group, buckets)
        self._reducer.broadcast_state()        # replicas start from rank 0's parameters (and Adam moments), like Lightning's DDP wrapper
        return self._reducer

    def _boundary(self, *mems: torch.Tensor):
        """Memory hand-off encoder(s) -> [mixer ->] decoder: when backward gets here every gradient of the decoder-side bucket
        is final."""
        if not torch.is_grad_enabled():
            return mems[0] if len(mems) == 1 else mems
        from .ddp import GradBoundary      # without a reducer the node only starts the decoder's collected weight gradients (runtime.WgradStream)
        return GradBoundary.apply(self._reducer, (1,), *mems)

    def configure_optimizers(self):
        """torch.optim.Adam(lr=1e-4, amsgrad=False) over all parameters (model.py:134-139,475-483) as ONE fused kernel."""
        return self.make_optimizer(lr=1e-4)

    def _i2w(self, token: int) -> str:
        d = self.i2w
        return d[token] if token in d else d[str(token)]

    @torch.no_grad()
    def _greedy(self, memory: torch.Tensor, want_probs: bool = False, use_cache: bool = True, chunk: int = 16):
        """Autoregressive loop of validation_step / get_pred_seq_and_pred_prob_seq (model.py:182-193,247-260):
        bs=1, memory_len=None, argmax of the last-step logits, stop after <eos> or max_seq_len tokens.
        use_cache=True runs the KV-cached native executor (Decoder.decode_tokens: `chunk` tokens per host call, the chosen
        token chained to the next position on the device; the host reads a chunk back at a time and cuts the sequence after
        <eos>, so at most chunk - 1 positions are computed in vain); use_cache=False re-runs the whole prefix each step
        exactly like the reference.  Both give the same tokens (tests/test_model_gpu.py)."""
        sos = self.w2i[SOS_TOKEN]
        tok = torch.full((1, 1), sos, dtype=torch.int64, device=memory.device)
        yhat: List[str] = []
        probs: List[float] = []
        if use_cache:
            state = self.decoder.init_decode(memory)
            left = self.max_seq_len
            while left > 0:
                n = min(chunk, left)
                toks, top1 = self.decoder.decode_tokens(tok, state, n)
                toks_h, top1_h = toks[:, 0].cpu().tolist(), (top1[:, 0].cpu().tolist() if want_probs else None)      # one sync per chunk
                for i, token in enumerate(toks_h):
                    word = self._i2w(token)
                    yhat.append(word)
                    if want_probs:
                        probs.append(float(top1_h[i]))
                    if word == EOS_TOKEN:
                        return yhat, probs
                tok = toks[-1].view(1, 1)
                left -= n
            return yhat, probs
        y_in = tok
        for _ in range(self.max_seq_len):
            logits = self.decoder(tgt=y_in, memory=memory, memory_len=None)   # [1, V, t]
            last = logits[0, :, -1]
            last32 = K.cast(last.contiguous(), torch.float32) if last.dtype != torch.float32 else last.contiguous()
            idx, val = K.argmax(last32)
            token = int(idx.item())
            word = self._i2w(token)
            yhat.append(word)
            if want_probs:
                probs.append(float(val.item()))
            if word == EOS_TOKEN:
                break
            y_in = torch.cat([y_in, idx.view(1, 1)], dim=1)
        return yhat, probs

    @torch.no_grad()
    def greedy_batch(self, memory: torch.Tensor, sync_every: int = 8) -> List[List[str]]:
        """KV-cached greedy decode of B same-sized inputs at once (the reference loops bs = 1,
        model.py:182-193).  Rows of the batch never interact, so each sequence equals what `_greedy` returns for that sample
        alone (tests/test_model_gpu.py).  Inputs must share their size: the reference pads nothing at inference, and padding
        would change the encoder features.  The host reads the chosen tokens back every `sync_every` steps only."""
        B = memory.shape[0]
        sos, eos = self.w2i[SOS_TOKEN], self.w2i[EOS_TOKEN]
        tok = torch.full((B, 1), sos, dtype=torch.int64, device=memory.device)
        state = self.decoder.init_decode(memory)
        done = [False] * B
        out: List[List[int]] = [[] for _ in range(B)]
        left = self.max_seq_len
        while left > 0 and not all(done):
            n = min(sync_every, left)
            toks, _ = self.decoder.decode_tokens(tok, state, n)      # n positions, tokens chained on the device
            for row in toks.cpu().tolist():                           # one device sync for the whole chunk
                for b, t in enumerate(row):
                    if not done[b]:
                        out[b].append(t)
                        done[b] = t == eos
            tok = toks[-1].view(B, 1)
            left -= n
        return [[self._i2w(t) for t in seq] for seq in out]

    @torch.no_grad()
    def beam_search(self, memory: torch.Tensor, beam: int = 4) -> Tuple[List[str], float]:
        """Beam-search decode of ONE input (BASELINE config C5; an extension -- the reference only decodes greedily).  The
        `beam` hypotheses are the batch rows of the KV-cached step; scores are sums of log-probabilities (no length
        normalisation); beam = 1 reproduces `_greedy` token for token.  Returns (words incl. <eos> if reached, score)."""
        assert memory.shape[0] == 1 and beam >= 1
        sos, eos = self.w2i[SOS_TOKEN], self.w2i[EOS_TOKEN]
        dev = memory.device
        state = self.decoder.init_decode(memory)
        state.share_memory_between(beam)                     # every hypothesis reads the same memory K|V; own self-attention cache rows
        tok = torch.full((beam, 1), sos, dtype=torch.int64, device=dev)
        scores = [0.0] + [float("-inf")] * (beam - 1)       # only the first row is a real hypothesis before the first step
        seqs: List[List[int]] = [[] for _ in range(beam)]
        best_done: Tuple[float, Optional[List[int]]] = (float("-inf"), None)
        exhausted = True                                     # the loop ran out of positions with hypotheses still alive
        for _ in range(self.max_seq_len):
            logits = self.decoder.decode_step(tok, state)
            idx, val = K.topk_logprob((logits if logits.dim() == 2 else logits.view(1, -1)).contiguous(), beam)
            idx_h, val_h = idx.cpu().tolist(), val.cpu().tolist()
            cands = [(scores[b] + val_h[b][j], b, idx_h[b][j]) for b in range(beam) if scores[b] > float("-inf") for j in range(beam)]
            cands.sort(key=lambda c: (-c[0], c[1], c[2]))
            parents, new_tok, new_scores, new_seqs = [], [], [], []
            for sc, b, t in cands:
                if t == eos:
                    if sc > best_done[0]:
                        best_done = (sc, seqs[b] + [t])
                    continue
                parents.append(b); new_tok.append(t); new_scores.append(sc); new_seqs.append(seqs[b] + [t])
                if len(parents) == beam:
                    break
            if not parents or new_scores[0] <= best_done[0]:        # scores only fall: no live hypothesis can overtake the best finished one
                exhausted = False
                break
            while len(parents) < beam:                                # pad with dead rows
                parents.append(parents[0]); new_tok.append(new_tok[0]); new_scores.append(float("-inf")); new_seqs.append([])
            pidx = torch.tensor(parents, dtype=torch.int64, device=dev)
            state.reorder_rows(pidx)
            tok = torch.tensor(new_tok, dtype=torch.int64, device=dev).view(beam, 1)
            scores, seqs = new_scores, new_seqs
        if exhausted and scores[0] > best_done[0]:
            best_done = (scores[0], seqs[0])                          # ran out of length: the best unfinished hypothesis wins
        return [self._i2w(t) for t in best_done[1]], best_done[0]

    @torch.no_grad()
    def test_step(self, batch, batch_idx) -> None:
        self.validation_step(batch, batch_idx)

    @torch.no_grad()
    def on_validation_epoch_end(self, name: str = "val", print_random_samples: bool = False) -> Dict[str, float]:
        # data-parallel evaluation: each rank decoded its shard; the edit-distance counts are all-reduced (metrics.py)
        metrics = compute_metrics_sharded(self.Y, self.YHat) if self._reducer is not None else compute_metrics(y_true=self.Y, y_pred=self.YHat)
        for k, v in metrics.items():
            self.log(f"{name}_{k}", v, prog_bar=True, logger=True, on_epoch=True)
        if print_random_samples:
            index = random.randint(0, len(self.Y) - 1)
            print(f"Ground truth - {self.Y[index]}")
            print(f"Prediction - {self.YHat[index]}")
        self.Y.clear()
        self.YHat.clear()
        return metrics

    @torch.no_grad()
    def on_test_epoch_end(self) -> Dict[str, float]:
        return self.on_validation_epoch_end(name="test", print_random_samples=True)


##################################################################### UNIMODAL TRANSFORMER


class Transformer(_Base):
    """model.py:54-262."""

    def __init__(self, max_input_height: int, max_input_width: int, max_seq_len: int, w2i: Dict[str, int], i2w: Dict[int, str],
                 ytest_i2w: Optional[Dict[int, str]] = None, attn_window: int = -1, teacher_forcing_prob: float = 0.5,
                 config: Optional[ModelConfig] = None) -> None:
        super().__init__()
        self.save_hyperparameters()
        self._common_init(w2i, i2w, ytest_i2w, max_seq_len, attn_window, teacher_forcing_prob, config)
        self.max_input_height, self.max_input_width = max_input_height, max_input_width
        c = self.config
        self.encoder = Encoder(in_channels=NUM_CHANNELS, dropout=c.encoder_dropout, out_channels=c.d_model)
        self.pos_2d = PositionalEncoding2D(num_channels=c.d_model, max_height=math.ceil(max_input_height / HEIGHT_REDUCTION),
                                           max_width=math.ceil(max_input_width / WIDTH_REDUCTION), dropout_p=c.dropout)
        self.decoder = self._make_decoder()

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """encoder -> 2-D PE -> flatten -> [B, S, d] (model.py:143-147,176-180)."""
        flat = self.ensure_flat()
        x = x.to(flat.device, non_blocking=True)
        f = self.encoder.forward_nhwc(x, flat.compute_dtype).permute(0, 3, 1, 2)
        return self._boundary(_flatten_memory(self.pos_2d(f)))

    def forward(self, x: torch.Tensor, xl: torch.Tensor, y_in: torch.Tensor) -> torch.Tensor:
        mem = self.encode(x)
        return self.decoder(tgt=y_in, memory=mem, memory_len=xl)     # host-resident ids / lengths go as they are: the decoder reads them before the copy

    def apply_teacher_forcing(self, y: torch.Tensor) -> torch.Tensor:
        """model.py:152-160: each non-pad token is replaced w.p. teacher_forcing_prob by randint(0, V-1) drawn
        from Python's `random` in row-major order (same draw sequence as the reference's double loop)."""
        out = y.detach().to("cpu", torch.int64).contiguous().clone()
        # The B x T double loop runs in C on the interpreter's OWN generator state (omr_teacher_forcing_noise: the calls
        # CPython's random.random / random.randint would make, in the reference's order), 1.8 ms -> 0.2 ms per C2 step.
        version, internal, gauss = random.getstate()
        state = (ctypes.c_uint * 624)(*internal[:624])
        index = ctypes.c_int(internal[624])
        lib().call("omr_teacher_forcing_noise", out.data_ptr(), out.numel(), float(self.teacher_forcing_prob), int(self.padding_idx), len(self.w2i),
                   ctypes.cast(state, ctypes.c_void_p), ctypes.byref(index))
        random.setstate((version, tuple(state) + (index.value,), gauss))
        out = out.to(y.dtype)
        return out.pin_memory() if (not y.is_cuda and torch.cuda.is_available()) else out.to(y.device)

    def training_step(self, batch, batch_idx) -> torch.Tensor:
        x, xl, y_in, y_out = batch
        y_in = self.apply_teacher_forcing(y_in)
        yhat = self.forward(x=x, xl=xl, y_in=y_in)
        loss = self.compute_loss(yhat, _h2d(y_out, yhat.device))
        self.log("train_loss", loss, prog_bar=True, logger=True, on_epoch=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx) -> None:
        x, y = batch
        assert x.size(0) == y.size(0) == 1, "Inference only supports batch_size = 1"
        yhat, _ = self._greedy(self.encode(x))
        self.Y.append([self.ytest_i2w[i.item()] for i in y[0][1:]])
        self.YHat.append(yhat)

    @torch.no_grad()
    def get_pred_seq_and_pred_prob_seq(self, x: torch.Tensor) -> Tuple[List[str], List[float]]:
        """model.py:226-262 (returns the top-1 LOGIT as "probability", like the reference)."""
        assert x.size(0) == 1, "Inference only supports batch_size = 1"
        return self._greedy(self.encode(x), want_probs=True)


##################################################################### MULTIMODAL TRANSFORMER


class CrossAttention(nn.Module):
    """model.py:268-355.  The attention-weights output of the reference (need_weights=True, discarded by every
    caller) is not materialised: the second return value is None."""

    def __init__(self, feature_dim: int, num_heads: int = 4, dropout: float = 0.1) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.attention = MultiheadAttention(feature_dim, num_heads, dropout)

    def forward(self, query, len_query, key_value, len_key_value):
        blk_lq = blk_lkv = None
        if len_query is not None and len_key_value is not None:
            # create_attention_mask (model.py:329-355): block [lq:, lkv:] masked, tiled with .repeat(num_heads,1,1)
            # -> head (b, h) uses sample (b*H + h) % B (quirk 2); the kernel indexes the length vectors that way.
            blk_lq = len_query.to(query.device, torch.int32).contiguous()
            blk_lkv = len_key_value.to(query.device, torch.int32).contiguous()
        kv = self.attention.project_kv(key_value.contiguous())
        out = self.attention.cross_attention(query.contiguous(), kv, None, self.training, blk_lq, blk_lkv)
        return out, None


class MultimodalTransformer(_Base):
    """model.py:358-726."""

    def __init__(self, max_img_height: int, max_img_width: int, max_audio_height: int, max_audio_width: int, max_seq_len: int,
                 w2i: Dict[str, int], i2w: Dict[int, str], ytest_i2w: Optional[Dict[int, str]] = None, mixer_type: str = "concat",
                 attn_window: int = -1, teacher_forcing_prob: float = 0.5, teacher_forcing_modality_prob: float = 0.5,
                 config: Optional[ModelConfig] = None) -> None:
        super().__init__()
        self.save_hyperparameters()
        self._common_init(w2i, i2w, ytest_i2w, max_seq_len, attn_window, teacher_forcing_prob, config)
        self.max_img_height, self.max_img_width = max_img_height, max_img_width
        self.max_audio_height, self.max_audio_width = max_audio_height, max_audio_width
        self.teacher_forcing_modality_prob = teacher_forcing_modality_prob
        c = self.config
        self.image_encoder = Encoder(in_channels=NUM_CHANNELS, dropout=c.encoder_dropout, out_channels=c.d_model)
        self.image_pos_2d = PositionalEncoding2D(c.d_model, math.ceil(max_img_height / HEIGHT_REDUCTION),
                                                 math.ceil(max_img_width / WIDTH_REDUCTION), c.dropout)
        self.audio_encoder = Encoder(in_channels=NUM_CHANNELS, dropout=c.encoder_dropout, out_channels=c.d_model)
        self.audio_pos_2d = PositionalEncoding2D(c.d_model, math.ceil(max_audio_height / HEIGHT_REDUCTION),
                                                 math.ceil(max_audio_width / WIDTH_REDUCTION), c.dropout)
        self.decoder = self._make_decoder()
        if mixer_type == "concat":
            self.mixer = self.mixer_concat
        elif mixer_type in ("attn_img", "attn_audio", "attn_both"):
            self.cross_attn = CrossAttention(feature_dim=c.d_model, num_heads=c.nhead, dropout=c.dropout)
            self.mixer = getattr(self, "mixer_" + mixer_type)
        else:
            raise ValueError(f"Invalid mixer type: {mixer_type}")

    def _encode(self, enc: Encoder, pos: PositionalEncoding2D, x: torch.Tensor) -> torch.Tensor:
        flat = self.ensure_flat()
        f = enc.forward_nhwc(_h2d(x, flat.device), flat.compute_dtype).permute(0, 3, 1, 2)
        return _flatten_memory(pos(f))

    def encoder_forward(self, xi, xa, xli=None, xla=None, apply_teacher_forcing_modality: bool = False):
        """model.py:485-522: BOTH encoders always run; one memory may then be returned alone."""
        xi = self._encode(self.image_encoder, self.image_pos_2d, xi)
        xa = self._encode(self.audio_encoder, self.audio_pos_2d, xa)
        xi, xa = self._boundary(xi, xa)
        self._touched = None           # which sub-modules get gradients this step: the optimizer skips the others (FusedAdam)
        if apply_teacher_forcing_modality:
            modality = self._draw_modality()
            if modality == "image":
                self._touched = ("image_encoder", "decoder")
                return xi, xli
            elif modality == "audio":
                self._touched = ("audio_encoder", "decoder")
                return xa, xla
            elif modality == "both":
                x, xl = self.mixer(xi=xi, xa=xa, xli=xli, xla=xla)
            else:
                raise ValueError(f"Invalid modality: {modality}")
        else:
            x, xl = self.mixer(xi=xi, xa=xa, xli=xli, xla=xla)
        return x, xl

    def forward(self, xi, xli, xa, xla, y_in, apply_teacher_forcing_modality: bool = False) -> torch.Tensor:
        x, xl = self.encoder_forward(xi=xi, xa=xa, xli=xli, xla=xla, apply_teacher_forcing_modality=apply_teacher_forcing_modality)
        return self.decoder(tgt=y_in, memory=x, memory_len=xl)

    def apply_teacher_forcing(self, y: torch.Tensor) -> torch.Tensor:
        """model.py:545-559 (vectorised, torch RNG)."""
        random_mask = torch.rand_like(y, dtype=torch.float) < self.teacher_forcing_prob
        combined = random_mask & (y != self.padding_idx)
        random_indices = torch.randint(0, len(self.w2i), y.shape, device=y.device)
        return torch.where(combined, random_indices, y)

    def apply_teacher_forcing_modality(self) -> str:
        """model.py:561-575."""
        if random.random() < self.teacher_forcing_modality_prob:
            return "image" if random.random() < 0.5 else "audio"
        return "both"

    def _draw_modality(self) -> str:
        """The modality decision of a training step.  Single process: the reference's draws from the global `random` stream.
        Data parallel: the global draws are still consumed (the stream stays where the reference's would be), but the decision
        comes from the reducer's shared generator, so every rank drops the same modality and FusedAdam skips the same
        sub-modules everywhere, however each rank seeded `random` (ddp.GradReducer.broadcast_state)."""
        modality = self.apply_teacher_forcing_modality()
        rng = getattr(self._reducer, "shared_rng", None)
        if rng is not None:
            modality = ("image" if rng.random() < 0.5 else "audio") if rng.random() < self.teacher_forcing_modality_prob else "both"
        return modality

    def training_step(self, batch, batch_idx) -> torch.Tensor:
        xi, xli, xa, xla, y_in, y_out = batch
        y_in = self.apply_teacher_forcing(y_in)
        yhat = self.forward(xi=xi, xli=xli, xa=xa, xla=xla, y_in=y_in, apply_teacher_forcing_modality=True)
        loss = self.compute_loss(yhat, _h2d(y_out, yhat.device))
        self.log("train_loss", loss, prog_bar=True, logger=True, on_epoch=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx) -> None:
        xi, xa, y = batch
        assert xi.size(0) == xa.size(0) == y.size(0) == 1, "Inference only supports batch_size = 1"
        x, _ = self.encoder_forward(xi=xi, xa=xa, xli=None, xla=None, apply_teacher_forcing_modality=False)
        yhat, _ = self._greedy(x)
        self.Y.append([self.ytest_i2w[i.item()] for i in y[0][1:]])
        self.YHat.append(yhat)

    ##### MODALITY MIXERS (model.py:644-726)

    @staticmethod
    def _len_mask(n: int, lens: torch.Tensor, device) -> torch.Tensor:
        return torch.arange(n, device=device).unsqueeze(0) >= lens.to(device).long().unsqueeze(1)

    def mixer_concat(self, xi, xa, xli=None, xla=None):
        x = torch.cat([xi, xa], dim=1)
        if xli is not None and xla is not None:
            # bool mask -> true -inf masking in the decoder (model.py:663-672)
            xl = torch.cat([self._len_mask(xi.shape[1], xli, xi.device), self._len_mask(xa.shape[1], xla, xa.device)], dim=1)
        else:
            xl = None
        return x, xl

    def mixer_attn_img(self, xi, xa, xli=None, xla=None):
        x, _ = self.cross_attn(query=xa, len_query=xla, key_value=xi, len_key_value=xli)
        return x, (xla if (xli is not None and xla is not None) else None)

    def mixer_attn_audio(self, xi, xa, xli=None, xla=None):
        x, _ = self.cross_attn(query=xi, len_query=xli, key_value=xa, len_key_value=xla)
        return x, (xli if (xli is not None and xla is not None) else None)

    def mixer_attn_both(self, xi, xa, xli=None, xla=None):
        # model.py:723-725: the second attention sees the ALREADY ATTENDED audio (variable shadowing, quirk 3)
        xa, xla = self.mixer_attn_img(xi=xi, xa=xa, xli=xli, xla=xla)
        xi, xli = self.mixer_attn_audio(xi=xi, xa=xa, xli=xli, xla=xla)
        return self.mixer_concat(xi=xi, xa=xa, xli=xli, xla=xla)
