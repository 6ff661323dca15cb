"""The model inside the mask-predict loop: the decoder side of the reference's NAR S2UT model
`NARS2UTTransformerModel` (research/TranSpeech/nar_transformer.py:569-976, task speech_to_speech_fasttranslate,
fairseq/tasks/nat_s2s_task.py:107-127) on the HIP engine, with the interface the research IterativeRefinementGenerator drives
(research/TranSpeech/iterative_refinement_generator.py:131-160; mirror: diffnorm_amd/iterative_refinement.py with
`speech_source=True`): forward_encoder / initialize_output_tokens / forward_decoder / regenerate_length_beam /
encoder.reorder_encoder_out.

Scope: `TransformerUnitDecoder.forward` (:321-420: token embedding + sinusoidal positions, pre-norm self-attention / encoder
attention / ReLU FFN layers, final LayerNorm, output projection), the length predictor (:436-480) and `forward_decoder`'s
mask-predict update (:791-842, one kernel: dn_cmlm_step).  The speech ENCODER (conv subsampler + transformer over fbank frames) is
out of scope: its output is a given tensor -- the `encoder` here only carries it and re-orders it.  Weights come from a state dict
under the reference decoder's parameter names (`decoder.*` of the model's checkpoint).
"""
import ctypes as C
import math
from typing import Dict, List, Optional

import torch

from . import _lib, packing
from .engine import _Engine, _dtype_code
from .iterative_refinement import DecoderOut, cmlm_update


def sinusoidal_table(num: int, dim: int, padding_idx: int) -> torch.Tensor:
    """SinusoidalPositionalEmbedding.get_embedding (fairseq/modules/sinusoidal_positional_embedding.py:36-58): [sin | cos], row
    padding_idx zeroed; built in fp32 like upstream."""
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float) * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num, dtype=torch.float).unsqueeze(1) * f.unsqueeze(0)
    tab = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(num, -1)
    tab[padding_idx, :] = 0
    return tab.contiguous()


def pack_nar(sd: Dict[str, torch.Tensor], dim: int, ffn: int, layers: int, vocab: int, max_pos: int, pad: int, dtype: int) -> List[torch.Tensor]:
    """Reference-named decoder state dict -> the tensor table of dn_nar_create (include/diffnorm_hip.h)."""
    g = lambda k: sd[k].float()
    mat = lambda w: packing._mat(w, dtype, cols=w.shape[1])  # rows -> multiple of 128, K = the (64-multiple) width itself
    stack = lambda items: torch.stack(items).contiguous()
    qkv_W, qkv_b, so_W, so_b, cq_W, cq_b, ckv_W, ckv_b, co_W, co_b, f1W, f1b, f2W, f2b, lng, lnb = ([] for _ in range(16))
    for l in range(layers):
        p = f"layers.{l}."
        sa, ea = p + "self_attn.", p + "encoder_attn."
        qkv_W.append(mat(torch.cat([g(sa + "q_proj.weight"), g(sa + "k_proj.weight"), g(sa + "v_proj.weight")])))
        qkv_b.append(torch.cat([g(sa + "q_proj.bias"), g(sa + "k_proj.bias"), g(sa + "v_proj.bias")]))
        so_W.append(mat(g(sa + "out_proj.weight"))); so_b.append(g(sa + "out_proj.bias"))
        cq_W.append(mat(g(ea + "q_proj.weight"))); cq_b.append(g(ea + "q_proj.bias"))
        ckv_W.append(mat(torch.cat([g(ea + "k_proj.weight"), g(ea + "v_proj.weight")])))
        ckv_b.append(torch.cat([g(ea + "k_proj.bias"), g(ea + "v_proj.bias")]))
        co_W.append(mat(g(ea + "out_proj.weight"))); co_b.append(g(ea + "out_proj.bias"))
        f1W.append(mat(g(p + "fc1.weight"))); f1b.append(g(p + "fc1.bias"))
        f2W.append(mat(g(p + "fc2.weight"))); f2b.append(g(p + "fc2.bias"))
        lng.append(torch.stack([g(p + "self_attn_layer_norm.weight"), g(p + "encoder_attn_layer_norm.weight"), g(p + "final_layer_norm.weight")]))
        lnb.append(torch.stack([g(p + "self_attn_layer_norm.bias"), g(p + "encoder_attn_layer_norm.bias"), g(p + "final_layer_norm.bias")]))
    len_W = torch.zeros(256, dim)
    len_W[:] = g("embed_length.weight")
    return [g("embed_tokens.weight").contiguous(), sinusoidal_table(max_pos, dim, pad), packing._mat(len_W, _lib.DN_F32, cols=dim),
            stack(qkv_W), stack(qkv_b), stack(so_W), stack(so_b), stack(cq_W), stack(cq_b), stack(ckv_W), stack(ckv_b), stack(co_W), stack(co_b),
            stack(f1W), stack(f1b), stack(f2W), stack(f2b), stack(lng), stack(lnb), g("layer_norm.weight").contiguous(),
            g("layer_norm.bias").contiguous(), mat(g("output_projection.weight"))]


def pack_nar_encoder(sd: Dict[str, torch.Tensor], input_dim: int, conv_channels: int, kernel: int, dim: int, ffn: int, layers: int, max_pos: int,
                     pad: int, dtype: int) -> List[torch.Tensor]:
    """Reference-named encoder state dict (S2TTransformerEncoder.state_dict()) -> the tensor table of dn_nar_encoder_create."""
    g = lambda k: sd[k].float()
    mat = lambda w: packing._mat(w, dtype, cols=w.shape[1])
    stack = lambda items: torch.stack(items).contiguous()

    def conv_mat(w, cols=None):  # Conv1d weight [out, in, k] -> [out, k * in], column j * in + c (the gathered rows' order)
        w2 = w.permute(0, 2, 1).reshape(w.shape[0], -1)
        return packing._mat(w2, dtype, cols=cols or w2.shape[1])

    c0, c1 = g("subsample.conv_layers.0.weight"), g("subsample.conv_layers.1.weight")
    assert c0.shape == (conv_channels, input_dim, kernel) and c1.shape == (2 * dim, conv_channels // 2, kernel), (c0.shape, c1.shape)
    qkv_W, qkv_b, so_W, so_b, f1W, f1b, f2W, f2b, lng, lnb = ([] for _ in range(10))
    for l in range(layers):
        p = f"transformer_layers.{l}."
        sa = p + "self_attn."
        qkv_W.append(mat(torch.cat([g(sa + "q_proj.weight"), g(sa + "k_proj.weight"), g(sa + "v_proj.weight")])))
        qkv_b.append(torch.cat([g(sa + "q_proj.bias"), g(sa + "k_proj.bias"), g(sa + "v_proj.bias")]))
        so_W.append(mat(g(sa + "out_proj.weight"))); so_b.append(g(sa + "out_proj.bias"))
        f1W.append(mat(g(p + "fc1.weight"))); f1b.append(g(p + "fc1.bias"))
        f2W.append(mat(g(p + "fc2.weight"))); f2b.append(g(p + "fc2.bias"))
        lng.append(torch.stack([g(p + "self_attn_layer_norm.weight"), g(p + "final_layer_norm.weight")]))
        lnb.append(torch.stack([g(p + "self_attn_layer_norm.bias"), g(p + "final_layer_norm.bias")]))
    return [conv_mat(c0, cols=packing.padk(kernel * input_dim)), g("subsample.conv_layers.0.bias").contiguous(), conv_mat(c1),
            g("subsample.conv_layers.1.bias").contiguous(), sinusoidal_table(max_pos, dim, pad), stack(qkv_W), stack(qkv_b), stack(so_W), stack(so_b),
            stack(f1W), stack(f1b), stack(f2W), stack(f2b), stack(lng), stack(lnb), g("layer_norm.weight").contiguous(), g("layer_norm.bias").contiguous()]


class NarEncoderEngine(_Engine):
    """dn_nar_encoder_* of libdiffnorm_hip.so: the speech encoder of the NAR S2UT model (S2STransformerEncoder,
    research/TranSpeech/nar_transformer.py:40-76): one pass per utterance batch, in front of the refinement loop."""

    def __init__(self, state_dict, input_dim=80, conv_channels=1024, kernel=5, dim=512, ffn=2048, layers=12, heads=8, max_pos=6002, pad=1,
                 dtype="bf16", device="cuda:0"):
        self.dim, self.input_dim = dim, input_dim
        self.dtype = _dtype_code(dtype)
        super().__init__(device, pack_nar_encoder(state_dict, input_dim, conv_channels, kernel, dim, ffn, layers, max_pos, pad, self.dtype))
        cfg = _lib.NarEncConfig(input_dim, conv_channels, kernel, dim, ffn, layers, heads, max_pos, pad, self.dtype)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_nar_encoder_create(C.byref(cfg), self._table, len(self.tensors), C.byref(self.handle)), "dn_nar_encoder_create")

    def __del__(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.dn_nar_encoder_destroy(self.handle)
            self.handle = C.c_void_p()

    def forward(self, feats: torch.Tensor, src_lengths: torch.Tensor):
        """feats [B, L, input_dim], src_lengths [B] -> (enc_out fp32 [B, S, dim] batch-major, lengths int32 [B])."""
        B, L, Fd = feats.shape
        assert Fd == self.input_dim
        feats = feats.to(self.device, torch.float32).contiguous()
        sl = src_lengths.to(self.device, torch.int32).contiguous()
        S = int(self.lib.dn_nar_encoder_out_frames(L))
        out = torch.empty(B, S, self.dim, dtype=torch.float32, device=self.device)
        lens = torch.empty(B, dtype=torch.int32, device=self.device)
        wp, wn = self._aligned(self._workspace(int(self.lib.dn_nar_encoder_workspace_bytes(self.handle, B, L))))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_nar_encoder_forward(self.handle, feats.data_ptr(), sl.data_ptr(), B, L, out.data_ptr(), lens.data_ptr(), wp, wn,
                                                       _lib.current_stream()), "dn_nar_encoder_forward")
        return out, lens


class _HipEncoder:
    """`model.encoder` of the NAR S2UT model on the HIP engine: __call__(src_tokens [B, L, 80], src_lengths) -> the reference's encoder
    dict (encoder_out [S, B, D], encoder_padding_mask [B, S] -- an empty list when nothing is padded, s2t_transformer.py:364-373), with
    the batch-major copy and the subsampled lengths the decoder engine wants riding along; reorder_encoder_out as upstream (:387-420)."""

    def __init__(self, engine: "NarEncoderEngine"):
        self.engine = engine

    def __call__(self, src_tokens, src_lengths=None):
        x, lens = self.engine.forward(src_tokens, src_lengths)
        S = x.shape[1]
        pad = torch.arange(S, device=x.device)[None, :] >= lens[:, None]
        return {"encoder_out": [x.transpose(0, 1)], "encoder_padding_mask": [pad] if bool(pad.any()) else [], "encoder_embedding": [],
                "encoder_states": [], "src_tokens": [], "src_lengths": []}

    def reorder_encoder_out(self, enc, order):
        return _GivenEncoder.reorder_encoder_out(self, enc, order)


class NarDecoderEngine(_Engine):
    """dn_nar_* of libdiffnorm_hip.so: cross-attention keys / values once per batch, length prediction, one decoder pass."""

    def __init__(self, state_dict, dim=512, ffn=2048, layers=6, heads=8, vocab=1004, max_pos=1026, pad=1, dtype="bf16", device="cuda:0"):
        self.dim, self.ffn, self.layers, self.heads, self.vocab, self.pad = dim, ffn, layers, heads, vocab, pad
        self.dtype = _dtype_code(dtype)
        super().__init__(device, pack_nar(state_dict, dim, ffn, layers, vocab, max_pos, pad, self.dtype))
        cfg = _lib.NarConfig(dim, ffn, layers, heads, vocab, max_pos, pad, self.dtype)
        _lib.check(self.lib.dn_nar_create(C.byref(cfg), self._table, len(self.tensors), C.byref(self.handle)), "dn_nar_create")

    def __del__(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.dn_nar_destroy(self.handle)
            self.handle = None

    def _ws_for(self, B, T, S):
        return self._aligned(self._workspace(int(self.lib.dn_nar_workspace_bytes(self.handle, B, T, S))))

    def cross_kv(self, enc_out: torch.Tensor) -> torch.Tensor:
        """enc_out fp32 [B, S, D] -> the layers' encoder-attention keys / values [layers, B, S, 2 D] (engine dtype; fp32 for bf16x3)."""
        B, S, D = enc_out.shape
        assert D == self.dim and enc_out.is_contiguous() and enc_out.dtype == torch.float32 and enc_out.device == self.device
        tdt = torch.bfloat16 if self.dtype == _lib.DN_BF16 else torch.float16 if self.dtype == _lib.DN_F16 else torch.float32
        raw = torch.empty(int(self.lib.dn_nar_cross_kv_bytes(self.handle, B, S)) + 256, dtype=torch.uint8, device=self.device)
        off = (-raw.data_ptr()) % 256
        ckv = raw[off: off + self.layers * B * S * 2 * D * tdt.itemsize].view(tdt).view(self.layers, B, S, 2 * D)
        wp, wn = self._ws_for(B, 1, S)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_nar_cross_kv(self.handle, enc_out.data_ptr(), B, S, ckv.data_ptr(), wp, wn, _lib.current_stream()), "dn_nar_cross_kv")
        return ckv

    def predict_lengths(self, enc_out: torch.Tensor, src_lengths: torch.Tensor) -> torch.Tensor:
        B, S, _ = enc_out.shape
        out = torch.empty(B, dtype=torch.int32, device=self.device)
        sl = src_lengths.to(self.device, torch.int32).contiguous()
        wp, wn = self._ws_for(B, 1, S)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_nar_predict_lengths(self.handle, enc_out.data_ptr(), sl.data_ptr(), B, S, out.data_ptr(), wp, wn, _lib.current_stream()),
                       "dn_nar_predict_lengths")
        return out

    def forward(self, tokens: torch.Tensor, ckv: torch.Tensor, src_lengths: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tokens int32 [B, T] -> logits fp32 [B, T, vocab]."""
        B, T = tokens.shape
        S = ckv.shape[2]
        assert tokens.dtype == torch.int32 and tokens.is_contiguous() and ckv.is_contiguous() and ckv.shape[1] == B
        sl = src_lengths if (src_lengths.dtype == torch.int32 and src_lengths.device == self.device) else src_lengths.to(self.device, torch.int32)
        logits = torch.empty(B, T, self.vocab, dtype=torch.float32, device=self.device) if out is None else out
        assert logits.shape == (B, T, self.vocab) and logits.dtype == torch.float32 and logits.is_contiguous()
        wp, wn = self._ws_for(B, T, S)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dn_nar_decoder_forward(self.handle, tokens.data_ptr(), ckv.data_ptr(), sl.contiguous().data_ptr(), B, T, S, logits.data_ptr(),
                                                       wp, wn, _lib.current_stream()), "dn_nar_decoder_forward")
        return logits


class _GivenEncoder:
    """The speech encoder is out of this row's scope: `__call__` hands back the encoder output it is given (the reference's dict:
    encoder_out [S,B,D], encoder_padding_mask [B,S]); `reorder_encoder_out` follows S2TTransformerEncoder's
    (fairseq/models/speech_to_text/s2t_transformer.py:383-411) and also re-orders the cached encoder-attention keys / values."""

    def __call__(self, encoder_out, src_lengths=None):
        return encoder_out

    def reorder_encoder_out(self, enc, order):
        order = order.reshape(-1)
        out = {"encoder_out": [x.index_select(1, order) for x in enc["encoder_out"]],
               "encoder_padding_mask": [x.index_select(0, order) for x in enc["encoder_padding_mask"]],
               "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": []}
        for k in ("_ckv", "_src_len"):  # engine-side caches travel with the rows
            if k in enc:
                out[k] = enc[k].index_select(1 if k == "_ckv" else 0, order).contiguous()
        return out


class _RefineGraph:
    """ONE refinement iteration of the mask-predict loop -- the decoder pass (about 70 launches) and the CMLM update with the
    iteration index read from a device counter, then the counter's increment -- captured into a hipGraph (torch.cuda.CUDAGraph over
    this library's launches on the capture stream) over static buffers and replayed: an iteration is one graph launch instead of
    70 host launches.  Keyed by (B, T, S, max_step); the encoder-attention keys / values and source lengths are copied into the
    static buffers when the caller's tensors change (once per utterance batch, or after the generator re-ordered a shrunken batch)."""

    def __init__(self, model, B, T, S, max_step, ckv_like, slen_like):
        eng, dev = model.engine, model.device
        self.tokens = torch.empty(B, T, dtype=torch.int32, device=dev)
        self.scores = torch.empty(B, T, dtype=torch.float32, device=dev)
        self.predicted = torch.empty(B, T, dtype=torch.int32, device=dev)
        self.logits = torch.empty(B, T, eng.vocab, dtype=torch.float32, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ckv, self.slen = torch.empty_like(ckv_like), torch.empty_like(slen_like)
        self._src = (None, None)
        self.max_step, self.model, self.graph = max_step, model, None

    def bind(self, ckv, slen):
        key = (ckv.data_ptr(), slen.data_ptr(), ckv._version, slen._version)
        if key != self._src:
            self.ckv.copy_(ckv)
            self.slen.copy_(slen)
            self._src = key

    def _iteration(self):
        m, eng = self.model, self.model.engine
        eng.forward(self.tokens, self.ckv, self.slen, out=self.logits)
        B, T, V = self.logits.shape
        _lib.check(eng.lib.dn_cmlm_step_dev(self.logits.data_ptr(), self.tokens.data_ptr(), self.scores.data_ptr(), self.predicted.data_ptr(), B, T, V,
                                            self.step.data_ptr(), int(self.max_step), int(m.unk), int(m.pad), _lib.current_stream()), "dn_cmlm_step_dev")
        self.step.add_(1)

    def run(self, n: int = 1):
        """n iterations from the state in the static buffers (tokens, scores, step)."""
        with torch.cuda.device(self.model.device):
            if self.graph is None:
                self._iteration()  # eager first: workspaces and kernel attributes settle outside capture
                n -= 1
                cur = torch.cuda.current_stream()
                side = torch.cuda.Stream(device=self.model.device)
                side.wait_stream(cur)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side):
                        self._iteration()
                cur.wait_stream(side)
                self.graph = g
            for _ in range(n):
                self.graph.replay()


class NARS2UTDecoderModel:
    """forward_decoder / initialize_output_tokens / regenerate_length_beam of NARS2UTTransformerModel (:791-912) over the HIP decoder.
    `use_graph` (default on): an iteration of `forward_decoder` runs as ONE captured hipGraph launch (_RefineGraph) instead of ~70
    host launches -- same kernels on the same inputs, bit-identical results; `refine(state, enc, n)` runs n iterations back to back
    with no host work in between (a generator without the adaptive early stop, and the benchmark)."""
    allow_length_beam = True

    def __init__(self, decoder_state_dict, dim=512, ffn=2048, layers=6, heads=8, vocab=1004, pad=1, unk=3, dtype="bf16", device="cuda:0",
                 use_graph: bool = True, encoder_state_dict=None, encoder_kw=None):
        """encoder_state_dict (S2TTransformerEncoder.state_dict() names; encoder_kw = NarEncoderEngine's sizes): with it `forward_encoder`
        runs the speech encoder on the HIP engine from [B, L, 80] features, as NARS2UTTransformerModel.forward_encoder does (:566-567);
        without it the encoder output is a tensor the caller supplies."""
        self.engine = NarDecoderEngine(decoder_state_dict, dim, ffn, layers, heads, vocab, pad=pad, dtype=dtype, device=device)
        if encoder_state_dict is not None:
            kw = dict(dim=dim, heads=heads, dtype=dtype, device=device)
            kw.update(encoder_kw or {})
            self.encoder = _HipEncoder(NarEncoderEngine(encoder_state_dict, **kw))
        else:
            self.encoder = _GivenEncoder()
        self.pad, self.unk, self.device = pad, unk, self.engine.device
        self.use_graph, self._graphs = use_graph, {}

    def _graph_for(self, B, T, ckv, slen, max_step) -> "_RefineGraph":
        key = (B, T, ckv.shape[2], int(max_step))
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 8:  # a generator whose batch keeps shrinking: keep the cache bounded
                self._graphs.pop(next(iter(self._graphs)))
            g = self._graphs[key] = _RefineGraph(self, B, T, ckv.shape[2], max_step, ckv, slen)
        g.bind(ckv, slen)
        return g

    def refine(self, decoder_out, encoder_out, n_iters: int):
        """n_iters mask-predict iterations from `decoder_out` (its `step` / `max_step` as the generator sets them) with no host work
        in between: the captured iteration replayed while the device step counter advances.  -> the DecoderOut after them."""
        ckv, slen = self._prepared(encoder_out)
        B, T = decoder_out.output_tokens.shape
        g = self._graph_for(B, T, ckv, slen, decoder_out.max_step)
        g.tokens.copy_(decoder_out.output_tokens)
        g.scores.copy_(decoder_out.output_scores)
        g.step.fill_(int(decoder_out.step))
        g.run(n_iters)
        return decoder_out._replace(output_tokens=g.tokens.long(), output_scores=g.scores.clone(), attn=None, step=decoder_out.step + n_iters)

    def eval(self):
        return self

    def forward_encoder(self, encoder_inputs):
        return self.encoder(*encoder_inputs)

    def _prepared(self, enc):
        """The engine's view of an encoder output, cached in the dict: batch-major fp32 copy -> cross keys / values; source lengths."""
        if "_ckv" not in enc:
            x = enc["encoder_out"][0].to(self.device, torch.float32).transpose(0, 1).contiguous()  # [S,B,D] -> [B,S,D]
            S, B = enc["encoder_out"][0].shape[:2]
            pad = enc["encoder_padding_mask"][0] if len(enc["encoder_padding_mask"]) > 0 else None
            enc["_src_len"] = ((~pad).sum(1) if pad is not None else torch.full((B,), S)).to(self.device, torch.int32)
            enc["_ckv"] = self.engine.cross_kv(x)
            enc["_bsd"] = x
        return enc["_ckv"], enc["_src_len"]

    def initialize_output_tokens(self, encoder_out, src_lengths, true_length=None):
        """(:844-885): lengths from the predictor (or given), rows of `unk` padded with `pad`, zero scores."""
        self._prepared(encoder_out)
        if true_length is not None:
            lengths = true_length.to(self.device).long()
        else:
            lengths = self.engine.predict_lengths(encoder_out["_bsd"], encoder_out["_src_len"]).long()
        return DecoderOut(*self._blank(lengths), None, 0, 0, None)

    def _blank(self, lengths):
        lengths = lengths.clamp(min=2)
        idx = torch.arange(int(lengths.max()), device=lengths.device)
        tok = torch.full((lengths.size(0), idx.numel()), self.pad, dtype=torch.long, device=lengths.device)
        tok = tok.masked_fill(idx[None, :] < lengths[:, None], self.unk)
        return tok, torch.zeros(tok.shape, dtype=torch.float32, device=lengths.device)

    def regenerate_length_beam(self, decoder_out, beam_size):
        """(:887-912)."""
        lengths = decoder_out.output_tokens.ne(self.pad).sum(1)
        lengths = (lengths[:, None] + torch.arange(beam_size, device=lengths.device)[None, :] - beam_size // 2).view(-1)
        tok, sc = self._blank(lengths)
        return decoder_out._replace(output_tokens=tok, output_scores=sc)

    def forward_decoder(self, decoder_out, encoder_out, decoding_format=None, **kwargs):
        """(:791-842): one decoder pass (HIP engine) + the mask-predict update (dn_cmlm_step: arg-max of the log-softmax into the masked
        positions, then re-masking of the lowest-scoring ones unless this is the last iteration)."""
        ckv, slen = self._prepared(encoder_out)
        step, max_step, history = decoder_out.step, decoder_out.max_step, decoder_out.history
        if self.use_graph:
            B, T = decoder_out.output_tokens.shape
            g = self._graph_for(B, T, ckv, slen, max_step)
            g.tokens.copy_(decoder_out.output_tokens)
            g.scores.copy_(decoder_out.output_scores)
            g.step.fill_(int(step))
            g.run(1)
            if history is not None:
                history.append(g.predicted.long())
                if (step + 1) < max_step:
                    history.append(g.tokens.long())
            return decoder_out._replace(output_tokens=g.tokens.long(), output_scores=g.scores.clone(), attn=None, history=history)
        tokens = decoder_out.output_tokens.to(self.device, torch.int32).contiguous()
        scores = decoder_out.output_scores.to(self.device, torch.float32).contiguous().clone()
        logits = self.engine.forward(tokens, ckv, slen)
        predicted, tokens = cmlm_update(logits, tokens, scores, step, max_step, self.unk, self.pad)
        if history is not None:
            history.append(predicted.long())
            if (step + 1) < max_step:
                history.append(tokens.long().clone())
        return decoder_out._replace(output_tokens=tokens.long(), output_scores=scores, attn=None, history=history)
