"""Fused APLA training step for MI355X: explicit forward + backward + gradient exchange + optimizer over the HIP kernels.

This replaces, for the hot path only, what the reference runs through autograd/DDP/torch.optim
(defaults/trainer.py:106-151 -> Classifier.forward -> VisionTransformer.forward_features -> Block.forward ->
APLA_Attention.forward, plus DDP's bucketed all-reduce, clip_grad_norm_ and AdamW).  Design points:

* No autograd graph.  Under APLA every weight except {proj_weight1, proj_bias1} x L and the head is frozen, so the
  backward is a fixed dX chain plus L column-masked dW1 products and the head.  It is written out as a static launch
  sequence on one HIP stream and captured into hipGraphs (torch.cuda.CUDAGraph drives hipStreamBeginCapture).
* Engine weight layout (built once, 288 GB of HBM makes duplicates free): every frozen Linear is kept as bf16 [out,in]
  for the forward and as a transposed bf16 copy [in,out] for the dX backward, so that all dense work is the same
  "NT" MFMA kernel.  LayerScale (frozen) is folded into the following weight/bias in both directions.  The APLA
  projection is kept in NATURAL feature order: its r trainable rows are re-scattered from the fp32 masters once per
  step (apla_pack_proj_rows), which removes the two activation-side scatter_ calls of appla_attn.py:70-79.
* Activations bf16, accumulation fp32, residual stream fp32 (default) or bf16, LN statistics / softmax / LSE fp32.
  The residual stream is ONE buffer updated in place; gamma / beta of the frozen LayerNorms are folded into qkv and fc1 / w12, so
  the LayerNorm kernels write the normalised row itself.  Saved per block for backward: the two normalised rows (16-bit) and
  their rstd, qkv, attention output, LSE, gelu'(pre-activation) (or SwiGLU's x12).
* Trainable state is ONE flat fp32 buffer (params / grads / Adam moments) in ``named_parameters`` order; the module's
  Parameters are views into it, so ``state_dict()`` and checkpoints keep the reference layout.
* Data parallel: one process per GPU; after backward the flat gradient buffer is all-reduced (RCCL, SUM) on a side
  stream in two chunks following backward order (head + upper blocks first, overlapped with the rest of backward);
  the mean (1/world) is folded into the fused clip+AdamW kernel.  No per-step barrier, no buffer broadcast
  (SURVEY.md §2.3 C1/C2).
"""
from dataclasses import dataclass
from typing import List, Optional

import os

import torch
import torch.nn as nn

from . import ops
from ._lib import AplaHipError
from .apla.appla_attn import APLA_Attention
from .dist import GradExchanger



@dataclass
class OptimConfig:
    lr: float = 1e-4
    weight_decay: float = 1e-5
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    grad_clipping: float = 1.0   # defaults/trainer.py:130 (0 disables)


class _BlockState:
    pass


def _bf(t):
    return t.detach().to(ops.half()).contiguous()


def _bf_t(t):
    return t.detach().t().contiguous().to(ops.half())


def _interleave_rows(w):
    h = w.shape[0] // 2
    return torch.stack([w[:h], w[h:]], 1).reshape(w.shape).contiguous()


def _half_mode(fn):
    """Run a public engine method against the library built for the engine's operand dtype."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with ops.use_half(self.hdt):
            return fn(self, *a, **k)
    return wrapped


class AplaTrainEngine:
    def __init__(self, model: nn.Module, batch_size: int, img_size: int, device="cuda", res_dtype=torch.float32,
                 grad_dtype=None, optim: Optional[OptimConfig] = None, process_group=None, use_graphs: bool = True,
                 compute_dtype=torch.bfloat16, loss_scale: float = 1.0, soft_targets: bool = False):
        if not torch.cuda.is_available():
            raise AplaHipError("AplaTrainEngine needs an MI355X (no CPU fallback)")
        self.device = torch.device(device)
        self.model = model.to(self.device)
        bb = model.backbone
        self.B, self.S = batch_size, img_size
        self.patch = bb.patch_size
        self.D, self.L, self.H = bb.embed_dim, bb.depth, bb.num_heads
        if self.D != 64 * self.H:
            raise NotImplementedError("HIP attention kernel needs head_dim == 64")
        # Element-wise dropout (main.py --dr / --adr; 0 in every shipped configuration): the nn.Dropout sites of the reference — pos_drop
        # (vit.py:395), attn_drop on the attention probabilities and proj_drop (appla_attn.py:58, :82), Mlp.drop after the activation and
        # after fc2 (vit.py:152-168) — run as counter-based mask passes around the step's launches (apla_dropout_fwd / _bwd,
        # apla_attn_fwd_dropout / _bwd_dropout; masks reproducible from one seed + the step count).  Such a step is launched eagerly (the
        # counters change per step) and runs the last block on every token; a nn.Dropout anywhere else is refused, never ignored.
        pd = lambda m: float(m.p) if isinstance(m, nn.Dropout) else 0.0     # noqa: E731
        self.p_pos = pd(getattr(bb, "pos_drop", None))
        self.p_attn = [pd(getattr(blk.attn, "attn_drop", None)) for blk in bb.blocks]
        self.p_proj = [pd(getattr(blk.attn, "proj_drop", None)) for blk in bb.blocks]
        self.p_mlp = [pd(getattr(blk.mlp, "drop", None)) for blk in bb.blocks]
        known = {id(m) for m in [getattr(bb, "pos_drop", None)] + [getattr(blk.attn, a, None) for blk in bb.blocks for a in ("attn_drop", "proj_drop")]
                 + [getattr(blk.mlp, "drop", None) for blk in bb.blocks]}
        for name, mod in model.named_modules():
            if isinstance(mod, nn.Dropout) and mod.p > 0.0 and id(mod) not in known:
                raise NotImplementedError(f"{name}: this nn.Dropout is not one of the reference's sites (pos_drop, attn_drop, proj_drop, Mlp.drop); "
                                          "train on the module path (apla_amd.module_trainer)")
        allp = [self.p_pos] + self.p_attn + self.p_proj + self.p_mlp
        if any(not (0.0 <= q < 1.0) for q in allp):
            raise ValueError("dropout probabilities must be in [0, 1)")
        self.drop_on = any(q > 0.0 for q in allp)
        self._drop_seed = int(torch.empty((), dtype=torch.int64).random_()) if self.drop_on else 0   # torch.manual_seed repeats a run
        self._drop_step = 0
        # stochastic depth (vit.py:74-93, :257, :284-285; main.py --dpr): fused into the LayerNorm kernels as one factor per sample and
        # branch (apla_layernorm_fwd_dp / _bwd_dp), drawn per step into `dp_scale` BEFORE the captured launches run
        self.dp_rates = [float(getattr(getattr(blk, "drop_path", None), "drop_prob", 0.0) or 0.0) for blk in bb.blocks]
        if any(not (0.0 <= q < 1.0) for q in self.dp_rates):
            raise ValueError(f"drop_path rates must be in [0, 1): {self.dp_rates}")
        self.dp_on = any(q > 0.0 for q in self.dp_rates)
        self._dp_given = None
        self.Np = (img_size // self.patch) ** 2
        self.N = self.Np + 1
        self.M = self.B * self.N
        self.eps = bb.eps
        self.swiglu = bb.use_swiglu
        if compute_dtype not in (torch.bfloat16, torch.float16):
            raise TypeError("compute_dtype must be torch.bfloat16 (default) or torch.float16")
        self.hdt = compute_dtype                  # 16-bit operand type: selects libapla_hip.so / libapla_hip_f16.so
        # probability targets [B, C] instead of class ids (the reference's advanced_aug: Mixup / CutMix / label smoothing)
        self.soft_targets = soft_targets
        # loss scale for fp16 gradients: a float (static) or "dynamic" = GradScaler semantics kept on the device
        self.dynamic_scale = loss_scale == "dynamic"
        self.loss_scale = 1.0 if self.dynamic_scale else float(loss_scale)
        self.scaler = None   # device float32[8], see ops.adamw_step_dynamic
        self._scaler_calls = 0
        self._scaler_step0 = 0.0   # the scaler's "steps taken" slot when _scaler_calls was last reset (checkpoint.load_session)
        self.res_dtype, self.grad_dtype = res_dtype, (grad_dtype or compute_dtype)
        # Element-wise dropout runs INSIDE the step's kernels where it can (round 6, second form): proj_drop and the dropout after fc2 in
        # the LayerNorm that adds the branch (forward) and in the LayerNorm backward that produces the branch's incoming gradient
        # (a masked second copy), Mlp.drop after the activation in fc1's GELU epilogue, pos_drop as one pass — every mask from
        # {seed, step} in DEVICE memory, so the step stays a captured launch sequence.  Needs the 16-bit gradient stream (the masked
        # copy exists on that LayerNorm backward only); attention-probability dropout keeps its own kernels and an eager launch sequence.
        self.drop_fused = self.drop_on and self.grad_dtype == compute_dtype and os.environ.get("APLA_DROPOUT_PASSES") != "1"
        self.optim = optim or OptimConfig()
        # process_group: a group, None = the default group when torch.distributed is initialised with more than one rank, or False =
        # this engine never exchanges (a rank-local engine inside a multi-rank job: bench.py's parity checks on rank 0)
        self.local_only = process_group is False
        self.pg = None if self.local_only else process_group
        self.world = 1   # set from the GradExchanger below: ONE source for "how many ranks sum into the gradient buffer"
        self.use_graphs = use_graphs and (not self.drop_on or (self.drop_fused and not any(q > 0 for q in self.p_attn)))
        # diagnostic switch: APLA_FULL_LAST_BLOCK=1 runs the last block's forward on all rows (A/B of the CLS-only tail)
        self.cls_only_tail = os.environ.get("APLA_FULL_LAST_BLOCK") != "1" and not self.drop_on
        self.scale = bb.blocks[0].attn.scale
        self.step_count = 0
        self._graphs = None
        self._pack_batched = None
        self._fc1_events = None
        self._small_ws = {}
        self.cls_small_gemm = os.environ.get("APLA_NO_SMALL_GEMM") != "1"   # diagnostic switch for A/B timing
        if self.dynamic_scale:
            self.scaler = ops.new_scaler_state(self.device)
        with ops.use_half(self.hdt):
            self._build_flat_params()
            self._build_frozen_layout()
            self._alloc_buffers()
            self.refresh_weights()

    # ------------------------------------------------------------------ trainable state
    def _build_flat_params(self):
        named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
        expect = []
        self.blocks_train = []
        for i, blk in enumerate(self.model.backbone.blocks):
            a = blk.attn
            if isinstance(a, APLA_Attention):
                W1, b1, inds, r = a.proj_weight1, a.proj_bias1, a.inds, a.partial_size
                names = (f"backbone.blocks.{i}.attn.proj_weight1", f"backbone.blocks.{i}.attn.proj_bias1")
            else:  # partial_size: full (apla_vit.py:66-75): full-rank projection, identity permutation
                W1, b1, r = a.proj.weight, a.proj.bias, self.D
                inds = torch.arange(self.D)
                names = (f"backbone.blocks.{i}.attn.proj.weight", f"backbone.blocks.{i}.attn.proj.bias")
            if self.D % 64 != 0:       # (= 64 * heads anyway; widths that are not multiples of 128 — vit_tiny's 192 — run the 64-wide GEMM tile)
                raise NotImplementedError(f"engine needs dim % 64 == 0 (D={self.D})")
            expect += list(names)
            self.blocks_train.append((W1, b1, inds, r))
        expect += ["fc.weight", "fc.bias"]
        if [n for n, _ in named] != expect:
            raise NotImplementedError("engine supports exactly the APLA trainable set (proj rows of every block + head); "
                                      f"got {[n for n, _ in named][:6]}…")
        n_total = sum(p.numel() for _, p in named)
        dev = self.device
        self.flat_params = torch.empty(n_total, device=dev, dtype=torch.float32)
        self.flat_grads = torch.zeros(n_total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n_total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n_total, device=dev, dtype=torch.float32)
        self.decay_mask = torch.zeros(n_total, device=dev, dtype=torch.uint8)
        self.norm_ws = torch.zeros(512, device=dev, dtype=torch.float32)
        self.slices = {}
        off = 0
        for n, p in named:
            k = p.numel()
            self.flat_params[off:off + k].copy_(p.detach().reshape(-1).float())
            p.data = self.flat_params[off:off + k].view(p.shape)      # module params become views of the flat buffer
            p.grad = self.flat_grads[off:off + k].view(p.shape)
            if not (n.endswith(".bias") or p.ndim == 1):             # defaults/wrappers.py:205-221
                self.decay_mask[off:off + k] = 1
            self.slices[n] = (off, k, tuple(p.shape))
            off += k
        self.n_trainable = n_total
        self.names = [n for n, _ in named]
        # The backward is cut into segments at block boundaries; the gradients a segment produced are all-reduced (one chunk
        # of the flat buffer each, in backward order) while the next segment runs.  Only the last chunk's exchange is exposed,
        # so with world > 1 the backward is cut in four (the exposed chunk is block 0 alone); a single
        # process keeps two segments (nothing to overlap, fewer graph launches).
        # APLA_FORCE_EXCHANGE=1 (diagnostic): take the world > 1 path — four segments, collectives on the side stream — in a
        # process group of ONE rank, so that a single-GPU box exercises the real RCCL calls between the graph replays
        force = os.environ.get("APLA_FORCE_EXCHANGE") == "1" and self.pg is not None
        self.world = 1 if self.local_only else GradExchanger.world_of(self.pg)
        n_seg = min(4 if (self.world > 1 or force) else 2, max(self.L, 1))
        # segment s ends after the backward of block cut[s]; cut[-1] = 0.  e.g. L = 12, two segments: [6, 0]
        self.seg_cuts = [(self.L * (n_seg - 1 - s)) // n_seg for s in range(n_seg)]
        if n_seg == 4 and self.L >= 8:
            # exchanging: the LAST chunk is the one nothing overlaps, so it is made as small as possible — block 0 alone (its
            # backward is also the shortest: no attention part) — and the other three share the rest: L = 12 -> [8, 4, 1, 0]
            self.seg_cuts = [2 * self.L // 3, self.L // 3, 1, 0]
        # CUs the step's persistent GEMM launches leave to the RCCL kernels of the overlapped all-reduce (ops.reserved_cus; at
        # config 2 a reservation of 8 costs the GEMMs nothing — their last round has that many idle CUs anyway);
        # it matches the channel budget bench.py gives RCCL (NCCL_MAX_NCHANNELS)
        self.reserve_cus = int(os.environ.get("APLA_RESERVE_CUS", "8")) if (self.world > 1 or force) else 0
        bounds = [self.slices[self.names[2 * c]][0] for c in self.seg_cuts]       # start offset of block cut[s]'s tensors
        his = [n_total] + bounds[:-1]
        self.chunks = [(lo, hi) for lo, hi in zip(bounds, his)]
        assert self.chunks[-1][0] == 0 and all(hi > lo for lo, hi in self.chunks)
        self.exchanger = GradExchanger(self.flat_grads, self.chunks, self.pg, always=force, local=self.local_only)
        # the optimizer divides by exactly the number of ranks the exchanger sums over (process_group=None with an initialised
        # default group of N ranks exchanges over those N ranks: the 1/world of the DDP mean must follow)
        assert self.world == self.exchanger.world, (self.world, self.exchanger.world)

    def _grad_view(self, name):
        off, k, shape = self.slices[name]
        return self.flat_grads[off:off + k].view(shape)

    def _param_view(self, name):
        off, k, shape = self.slices[name]
        return self.flat_params[off:off + k].view(shape)

    # ------------------------------------------------------------------ frozen weights in kernel layout
    def _build_frozen_layout(self):
        bb, dev, D = self.model.backbone, self.device, self.D
        f32 = lambda t: t.detach().float().contiguous().to(dev)  # noqa: E731
        K = 3 * self.patch * self.patch
        self.Kp = (K + 63) // 64 * 64
        wpe = torch.zeros(D, self.Kp, device=dev)
        wpe[:, :K] = bb.patch_embed.proj.weight.detach().reshape(D, K).float()
        self.Wpe, self.bpe = wpe.to(ops.half()), f32(bb.patch_embed.proj.bias)
        self.cls = f32(bb.cls_token).reshape(D)
        with torch.no_grad():
            self.pos = f32(bb.interpolate_pos_encoding(self.Np)[0])  # bicubic resize once, not every forward
        self.gf, self.bf_ = f32(bb.norm.weight), f32(bb.norm.bias)
        self.blocks: List[_BlockState] = []
        for i, blk in enumerate(bb.blocks):
            st = _BlockState()
            W1p, b1p, inds, r = self.blocks_train[i]
            st.r = r
            # the dW kernel works on multiples of 64 rows: a partial size like 8 (the shipped configs) is padded with the next
            # entries of the block's index permutation (frozen features); their gradient rows are computed and dropped
            st.r_pad = (r + 63) // 64 * 64
            st.inds = inds.to(dev).int().contiguous()
            # gamma / beta of the two (frozen) LayerNorms are folded into the frozen Linear that reads the normalised row:
            # W' = W diag(gamma), b' = b + W beta.  The LayerNorm kernels then write xhat itself — which is also what the
            # backward needs, so the 16-bit xhat saved per LayerNorm replaces the fp32 residual rows (apla_layernorm_bwd_ex)
            g1, b1 = f32(blk.norm1.weight), f32(blk.norm1.bias)
            g2, b2 = f32(blk.norm2.weight), f32(blk.norm2.bias)
            fold = lambda w, b, g, be: (w * g[None, :], (b if b is not None else 0.0) + w @ be)  # noqa: E731
            a = blk.attn
            wq, st.bqkv = fold(f32(a.qkv.weight), f32(a.qkv.bias) if a.qkv.bias is not None else None, g1, b1)
            st.bqkv = st.bqkv.contiguous()
            st.Wqkv, st.WqkvT = _bf(wq), _bf_t(wq)
            gam1 = f32(blk.ls1.gamma) if hasattr(blk.ls1, "gamma") else None
            gam2 = f32(blk.ls2.gamma) if hasattr(blk.ls2, "gamma") else None
            st.gamma1 = gam1
            st.row_scale = gam1[st.inds[:st.r_pad].long()].contiguous() if gam1 is not None else None
            # natural-order merged projection; frozen rows written here, trainable rows by pack_proj_rows
            Wn = torch.zeros(D, D, device=dev)
            bn = torch.zeros(D, device=dev)
            if isinstance(a, APLA_Attention):
                fi = st.inds[r:].long()
                Wn[fi] = a.proj_weight2.detach().float()
                bn[fi] = a.proj_bias2.detach().float()
                if gam1 is not None:
                    Wn[fi] *= gam1[fi, None]
                    bn[fi] *= gam1[fi]
            if i == 0:  # one allocation for all blocks: the per-step re-scatter is a single launch when r is uniform
                L_ = len(bb.blocks)
                self.Wnat_all = torch.empty(L_, D, D, device=dev, dtype=ops.half())
                self.WnatT_all = torch.empty(L_, D, D, device=dev, dtype=ops.half())
                self.bnat_all = torch.empty(L_, D, device=dev)
            self.Wnat_all[i].copy_(Wn)
            self.WnatT_all[i].copy_(Wn.t())
            self.bnat_all[i].copy_(bn)
            st.Wnat, st.WnatT, st.bnat = self.Wnat_all[i], self.WnatT_all[i], self.bnat_all[i]
            mlp = blk.mlp
            if self.swiglu:
                st.F = mlp.w3.in_features
                w12, b12 = fold(f32(mlp.w12.weight), f32(mlp.w12.bias), g2, b2)
                w12 = _interleave_rows(w12)
                st.W12, st.W12T = w12.to(ops.half()), w12.t().contiguous().to(ops.half())
                st.b12 = _interleave_rows(b12).contiguous()
                w3, b3 = mlp.w3.weight.detach().float(), mlp.w3.bias.detach().float()
            else:
                st.F = mlp.fc1.out_features
                wf, st.bfc1 = fold(f32(mlp.fc1.weight), f32(mlp.fc1.bias), g2, b2)
                st.bfc1 = st.bfc1.contiguous()
                st.Wfc1, st.Wfc1T = _bf(wf), _bf_t(wf)
                w3, b3 = mlp.fc2.weight.detach().float(), mlp.fc2.bias.detach().float()
            if gam2 is not None:
                w3, b3 = w3 * gam2[:, None], b3 * gam2
            st.Wout, st.WoutT, st.bout = w3.to(ops.half()).contiguous(), w3.t().contiguous().to(ops.half()), b3.contiguous()
            st.W1_name, st.b1_name = self.names[2 * i], self.names[2 * i + 1]
            self.blocks.append(st)
        self.C = self.model.fc.out_features
        self._build_panel_images()

    def _build_panel_images(self):
        """K-panel images (ops.k_panels: [K/32, N, 32]) of the weights of the large plain-store GEMMs.  The fill path of a CU
        moves whole 128-byte lines; a 32-wide K-step of a row-major weight uses half of each (tools/dma_probe.hip), the image
        all of it: qkv / proj / fc2 / dfc1 / dproj / dqkv run 2-9 % faster, bit-identical.  Frozen weights are converted here
        once; the trainable rows of the merged projection are written into both layouts by the per-step pack kernel."""
        M, D, L = self.M, self.D, len(self.blocks)
        use = os.environ.get("APLA_W_PANELS", "1") != "0"
        def img(w, rows=M):
            return ops.k_panels(w.contiguous()) if use and ops.gemm_panel_ok(rows, w.shape[0], w.shape[1]) else w
        self.Wpe_i = img(self.Wpe, self.B * self.Np)
        nat = use and ops.gemm_panel_ok(M, D, D)
        self.Wnat_p_all = torch.empty(L, D // 32, D, 32, device=self.device, dtype=ops.half()) if nat else None
        self.WnatT_p_all = torch.empty(L, D // 32, D, 32, device=self.device, dtype=ops.half()) if nat else None
        for i, st in enumerate(self.blocks):
            st.Wqkv_i, st.WqkvT_i, st.Wout_i = img(st.Wqkv), img(st.WqkvT), img(st.Wout)
            st.Wkv_i = img(st.Wqkv[D:]) if i == L - 1 else None     # last block: K and V for every token, Q for the CLS rows only
            st.Wdfc1_i = img(st.W12T if self.swiglu else st.Wfc1T)
            # fc1 itself takes a weight image where it runs on the ping-pong kernel (the two-output GELU above 40 000 rows)
            st.Wfc1_i = None if self.swiglu else (ops.k_panels(st.Wfc1) if use and ops.gemm_panel_ok(M, st.Wfc1.shape[0], D, ops.EPI_GELU) else st.Wfc1)
            if nat:   # frozen rows now, trainable rows every step (refresh_weights)
                self.Wnat_p_all[i].copy_(ops.k_panels(st.Wnat))
                self.WnatT_p_all[i].copy_(ops.k_panels(st.WnatT))
                st.Wnat_i, st.WnatT_i = self.Wnat_p_all[i], self.WnatT_p_all[i]
            else:
                st.Wnat_i, st.WnatT_i = st.Wnat, st.WnatT

    # ------------------------------------------------------------------ activations / workspaces
    def _alloc_buffers(self):
        dev, M, D, B, N, H, L = self.device, self.M, self.D, self.B, self.N, self.H, self.L
        e = lambda *s, dt=ops.half(): torch.empty(*s, device=dev, dtype=dt)  # noqa: E731
        # stochastic depth: rows 2 i / 2 i + 1 = the per-sample factors floor(keep + u) / keep of block i's attention / MLP branch
        self.dp_scale = torch.ones(2 * L, B, device=dev, dtype=torch.float32) if self.dp_on else None
        if self.dp_on:
            self._dp_keep = torch.tensor([1.0 - q for q in self.dp_rates for _ in (0, 1)], device=dev, dtype=torch.float32)[:, None]
            self._dp_gen = torch.Generator(device=dev)
            self._dp_gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_()))     # one draw from torch's default CPU generator: torch.manual_seed repeats a run
        self.images = e(B, 3, self.S, self.S, dt=torch.float32)
        self.labels = torch.zeros(B, device=dev, dtype=torch.int32)
        self.targets = torch.zeros(B, self.C, device=dev, dtype=torch.float32) if self.soft_targets else None
        self.cols = e(B * self.Np, self.Kp)
        self.patches = e(B * self.Np, D)
        # ONE residual-stream buffer, updated in place by the LayerNorm kernels (x += branch fused): the backward reads the
        # normalised rows xh1 / xh2 (16-bit, one per LayerNorm) and rstd instead of the fp32 rows of every block
        self.res = e(M, D, dt=self.res_dtype)
        stat = lambda: e(M, dt=torch.float32)  # noqa: E731
        self.mean_scratch = stat()                  # the forward kernel's mean output: not needed again
        self.rstd1, self.rstd2 = [stat() for _ in range(L)], [stat() for _ in range(L)]
        self.xh1 = [e(M, D) for _ in range(L)]
        self.xh2 = [e(M, D) if (i < L - 1 or not self.cls_only_tail) else None for i in range(L)]  # last block: ln_cls (CLS rows only)
        self.qkv = [e(M, 3 * D) for _ in range(L)]
        self.o = [e(M, D) for _ in range(L)]
        self.lse = [e(B, H, N, dt=torch.float32) for _ in range(L)]
        Fsave = (2 * self.blocks[0].F) if self.swiglu else self.blocks[0].F
        self.act_saved = [e(M, Fsave) for _ in range(L)]       # gelu'(a) or interleaved x12
        self.branch = e(M, D)  # bf16 branch output (projection / fc2) awaiting the fused residual add
        self.h = e(M, self.blocks[0].F)
        # fc1's output h and dfc2's product go straight to a plain-store GEMM (fc2, dfc1): where both sides allow it they are
        # written as K-panel images by the producing epilogue (ops.gemm_nt with a 3-D `out`) and read as such
        # (element-wise dropout works on row-major h / GELU': its mask is indexed by the row-major element number, as the oracle's)
        F, use = self.blocks[0].F, os.environ.get("APLA_W_PANELS", "1") != "0" and not (self.drop_on and not self.drop_fused and any(q > 0 for q in self.p_mlp))
        epi_f, n_f = (ops.EPI_SWIGLU, 2 * F) if self.swiglu else (ops.EPI_GELU, F)       # fc1 / w12: epilogue and GEMM width
        self.h_img = use and ops.gemm_out_image_ok(M, n_f, D, epi_f) and ops.gemm_panel_ok(M, D, F)
        self.h_out = self.h.view(F // 32, M, 32) if self.h_img else self.h
        # gelu' (saved by fc1's epilogue for dfc2's) is private to those two epilogues: an image too, except in the last block,
        # whose CLS-only backward picks rows b*N out of the row-major buffer (SwiGLU saves x12 row-major)
        act_img = self.h_img and not self.swiglu and os.environ.get("APLA_ACT_IMG", "1") != "0"
        self.act_io = [a.view(F // 32, M, 32) if act_img and i < L - 1 else a for i, a in enumerate(self.act_saved)]
        # last block, forward: only the CLS row of every sequence is used downstream (final norm + x[:, 0])
        self.branch_cls = e(B, D)
        self.ln_cls = e(B, D)
        self.h_cls = e(B, self.blocks[0].F)
        self.mean2_cls, self.rstd2_cls = e(B, dt=torch.float32), e(B, dt=torch.float32)
        self.xn = e(B, D, dt=torch.float32)
        self.meanf, self.rstdf = e(B, dt=torch.float32), e(B, dt=torch.float32)
        self.logits = e(B, self.C, dt=torch.float32)
        self.dlogits = e(B, self.C, dt=torch.float32)
        self.row_loss = e(B, dt=torch.float32)
        self.loss = e(1, dt=torch.float32)
        self.dxn = e(B, D, dt=torch.float32)
        if self.drop_on:
            self.Gm = e(M, D)     # the incoming gradient of a branch through its dropout mask
            self.drop_rng = torch.zeros(2, device=dev, dtype=torch.int64)        # {seed, step}: what the in-kernel masks are drawn from
            self._sync_drop_rng()
        if self.drop_on and not self.drop_fused:    # pass form: keep bytes of the two branch sites per block, scratch for the others
            u8 = lambda n: torch.empty(n, device=dev, dtype=torch.uint8)       # noqa: E731
            self.keep_a = [u8(M * D) if self.p_proj[i] > 0 else None for i in range(L)]
            self.keep_b = [u8(M * D) if self.p_mlp[i] > 0 else None for i in range(L)]
            self.keep_h = u8(M * self.blocks[0].F) if any(q > 0 for q in self.p_mlp) else None
            self.keep_pos = u8(M * D) if self.p_pos > 0 else None
        # backward
        self.G = torch.zeros(M, D, device=dev, dtype=self.grad_dtype)               # residual-gradient stream (never re-zeroed)
        self.Gb = self.G if self.grad_dtype == ops.half() else torch.zeros(M, D, device=dev, dtype=ops.half())
        self.dact = e(M, Fsave)
        self.dact_img = use and ops.gemm_out_image_ok(M, F, D, ops.EPI_SWIGLU_BWD if self.swiglu else ops.EPI_MUL) and \
            ops.gemm_panel_ok(M, D, Fsave)
        self.dact_out = self.dact.view(Fsave // 32, M, 32) if self.dact_img else self.dact
        self.dln = e(M, D)
        self.dO = e(M, D)
        self.dqkv = e(M, 3 * D)
        self.delta = e(B, H, N, dt=torch.float32)
        rmax = max(st.r_pad for st in self.blocks)
        self.dW_tmp = e(rmax, D, dt=torch.float32) if any(st.r_pad != st.r for st in self.blocks) else None
        self.db_tmp = e(rmax, dt=torch.float32) if self.dW_tmp is not None else None
        # CLS-only backward of the last block (compact [B, .] operands)
        self.dact_cls, self.dln_cls, self.dO_cls = e(B, Fsave), e(B, D), e(B, D)
        self.dyg_cls = e(B * rmax)
        # dW1 / db1 of several blocks in one launch pair (apla_proj_dw_batched): the gathered gradient columns of a block are
        # kept (one [M, r] buffer per block instead of one for all) and the blocks of a backward segment are flushed
        # together, at most DW_BATCH at a time.  Needs one r for all blocks and no row padding; else block by block.
        self.dw_batch = 0
        if os.environ.get("APLA_DW_BATCH", "1") != "0" and len({st.r for st in self.blocks}) == 1 and \
                all(st.r_pad == st.r for st in self.blocks) and self.L > 2 and self.D % 128 == 0 and not self.drop_on:
            self.dw_batch = min(int(os.environ.get("APLA_DW_BATCH_MAX", "6")), ops.DW_MAX_BATCH)
        self._dw_pending = []
        if self.dw_batch > 1:
            self.dyg_all = e(self.L - 1, M * rmax)
            self.dw_ws_batched = {}   # layers in the batch -> workspace (sized by the C-ABI for that count)
        self.dyg = e(M * rmax)
        self.dw_ws = ops.dw_workspace(M, rmax, D, dev)
        for st in self.blocks:
            if ops.lib().apla_dw_workspace_bytes(M, st.r_pad, (D + 127) // 128 * 128) > self.dw_ws.numel() * 4:
                self.dw_ws = ops.dw_workspace(M, st.r_pad, D, dev)

    @_half_mode
    def refresh_frozen_copies(self):
        """Rebuild the kernel-layout copies of the frozen weights (bf16, transposed, LayerScale folded) after the module's
        frozen parameters were overwritten, e.g. by checkpoint.load_session.  Only before the step graphs are captured:
        the captured launches hold the old buffers' addresses."""
        if self._graphs is not None:
            raise RuntimeError("frozen weights cannot be replaced after hipGraph capture; load the checkpoint before the first step")
        self._build_frozen_layout()
        self._pack_batched = None  # LayerScale gammas may have changed

    # ------------------------------------------------------------------ step pieces
    @_half_mode
    def refresh_weights(self):
        """Re-scatter the trainable projection rows (fp32 masters -> natural-order bf16 weight, its transpose, bias)."""
        if self._pack_batched is None:  # uniform partial size and back-to-back (W1, b1) pairs in the flat buffer?
            rs = {st.r for st in self.blocks}
            offs = [self.slices[st.W1_name][0] for st in self.blocks]
            stride = self.blocks[0].r * self.D + self.blocks[0].r
            ok = len(rs) == 1 and all(o == offs[0] + k * stride for k, o in enumerate(offs)) and \
                all(self.slices[st.b1_name][0] == self.slices[st.W1_name][0] + st.r * self.D for st in self.blocks) and \
                len({st.gamma1 is None for st in self.blocks}) == 1
            if ok:
                self._inds_all = torch.stack([st.inds for st in self.blocks]).contiguous()
                self._gamma_all = None if self.blocks[0].gamma1 is None else torch.stack([st.gamma1 for st in self.blocks]).contiguous()
                self._pack_batched = (offs[0], stride)
            else:
                self._pack_batched = False
                for st in self.blocks:   # block-by-block packing keeps the row-major copies only
                    st.Wnat_i, st.WnatT_i = st.Wnat, st.WnatT
        if self._pack_batched:
            off0, stride = self._pack_batched
            ops.pack_proj_rows_batched(self.flat_params[off0:], stride, self._inds_all, self._gamma_all, self.Wnat_all,
                                       self.WnatT_all, self.bnat_all, self.blocks[0].r, self.Wnat_p_all, self.WnatT_p_all)
            return
        for st in self.blocks:
            ops.pack_proj_rows(self._param_view(st.W1_name), self._param_view(st.b1_name), st.inds, st.gamma1,
                               st.Wnat, st.WnatT, st.bnat)

    def _forward(self, inference: bool = False):
        """``inference``: the no-grad forward of forward_only — fc1 runs the forward-only GELU epilogue (GELU' is neither
        computed nor stored); everything else is the training forward."""
        B, N, H, D = self.B, self.N, self.H, self.D
        dps = (lambda k: None) if (inference or not self.dp_on) else (lambda k: self.dp_scale[k])     # stochastic depth: training only
        ops.patchify(self.images, self.patch, self.Kp, out=self.cols)
        ops.gemm_nt(self.cols, self.Wpe_i, self.bpe, out=self.patches, tag=ops.TAG_PATCH)
        ops.assemble_tokens(self.patches, self.cls, self.pos, B, self.Np, out=self.res)
        drop = self.drop_on and not inference
        fused = drop and self.drop_fused
        dsite = lambda site, p_: (self.drop_rng, self._drop_stride, site, p_)       # noqa: E731  (in-kernel mask of a site)
        if drop and self.p_pos > 0:       # pos_drop (vit.py:395)
            if fused:
                ops.dropout_dev(self.res, self.p_pos, self.drop_rng, self._drop_stride, 0, out=self.res)
            else:
                ops.dropout_fwd(self.res, self.p_pos, self._drop_seed, self._drop_offset(0), out=self.res, keep=self.keep_pos)
        # The residual updates x += branch (vit.py:284-285) are fused into the NEXT LayerNorm: the projection / fc2 GEMMs store
        # their (LayerScale-folded) branch output in bf16 — as the reference's fp16-autocast Linear does before the fp32
        # residual add — and the LN kernel forms x_new = x + branch in registers, writes it and normalises it in one pass.
        # The GEMM epilogues stay operand-free (eligible for the ping-pong kernel), the adds cost no kernel of their own.
        for i, st in enumerate(self.blocks):
            xh1 = self.xh1[i]
            if i == 0:
                ops.layernorm_fwd(self.res, None, None, self.eps, out=xh1, mean=self.mean_scratch, rstd=self.rstd1[0])
            else:
                ops.layernorm_fwd(self.res, None, None, self.eps, out=xh1, mean=self.mean_scratch, rstd=self.rstd1[i],
                                  add=self.branch, x_out=self.res, add_scale=dps(2 * i - 1), scale_period=N,
                                  drop=dsite(3 + 4 * (i - 1), self.p_mlp[i - 1]) + (D,) if fused and self.p_mlp[i - 1] > 0 and not self.swiglu else None)
            if i == self.L - 1 and self.cls_only_tail:
                # last block: K and V for every token, Q for the CLS rows only (the only query that is ever used)
                ops.gemm_nt(xh1, st.Wkv_i, st.bqkv[D:], out=self.qkv[i][:, D:], tag=ops.TAG_QKV)
                self._gemm_rows(xh1.view(B, N * D)[:, :D], st.Wqkv[:D], st.bqkv[:D], out=self.qkv[i].view(B, N * 3 * D)[:, :D])
                self._forward_last_block_tail(st, i, dps)
                break
            ops.gemm_nt(xh1, st.Wqkv_i, st.bqkv, out=self.qkv[i], tag=ops.TAG_QKV)
            if drop and self.p_attn[i] > 0:   # attn_drop (appla_attn.py:58): the mask is rebuilt by the backward from the same counters
                ops.attn_fwd_dropout(self.qkv[i], B, N, H, self.scale, self.p_attn[i], self._drop_seed, self._drop_offset(4 + 4 * i),
                                     o=self.o[i], lse=self.lse[i])
            else:
                ops.attn_fwd(self.qkv[i], B, N, H, self.scale, o=self.o[i], lse=self.lse[i])
            ops.gemm_nt(self.o[i], st.Wnat_i, st.bnat, out=self.branch, tag=ops.TAG_PROJ)
            if drop and not fused and self.p_proj[i] > 0:   # proj_drop (appla_attn.py:82; commutes with the LayerScale vector folded into the weight)
                ops.dropout_fwd(self.branch, self.p_proj[i], self._drop_seed, self._drop_offset(1 + 4 * i), out=self.branch, keep=self.keep_a[i])
            xh2 = self.xh2[i]
            ops.layernorm_fwd(self.res, None, None, self.eps, out=xh2, mean=self.mean_scratch, rstd=self.rstd2[i],
                              add=self.branch, x_out=self.res, add_scale=dps(2 * i), scale_period=N,
                              drop=dsite(1 + 4 * i, self.p_proj[i]) + (D,) if fused and self.p_proj[i] > 0 else None)
            ev = self._fc1_events
            if ev is not None:   # bench.py: HIP events around the dominant launch, in its place inside the step (eager replay only)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.swiglu:
                ops.gemm_nt(xh2, st.W12, st.b12, epilogue=ops.EPI_SWIGLU, aux_out=self.act_saved[i], out=self.h_out)
            elif inference:
                ops.gemm_nt(xh2, st.Wfc1, st.bfc1, epilogue=ops.EPI_GELU_FWD, out=self.h_out)
            elif fused and self.p_mlp[i] > 0:      # Mlp.drop after the activation inside the epilogue: h and GELU' through the same mask
                ops.gemm_nt(xh2, st.Wfc1, st.bfc1, epilogue=ops.EPI_GELU, aux_out=self.act_io[i], out=self.h_out, drop=dsite(2 + 4 * i, self.p_mlp[i]))
            else:
                ops.gemm_nt(xh2, st.Wfc1_i, st.bfc1, epilogue=ops.EPI_GELU, aux_out=self.act_io[i], out=self.h_out)
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
            mlp_drop = drop and not fused and self.p_mlp[i] > 0 and not self.swiglu        # Mlp.drop, both sites (vit.py:164-167); SwiGLUFFNFused has none
            if mlp_drop:    # after the activation: h is masked, and GELU' with the same mask — dfc2's epilogue then needs no change
                ops.dropout_fwd(self.h, self.p_mlp[i], self._drop_seed, self._drop_offset(2 + 4 * i), out=self.h, keep=self.keep_h)
                ops.dropout_bwd(self.act_saved[i], self.keep_h, self.p_mlp[i], out=self.act_saved[i])
            ops.gemm_nt(self.h_out, st.Wout_i, st.bout, out=self.branch, tag=ops.TAG_FC2)
            if mlp_drop:    # after fc2
                ops.dropout_fwd(self.branch, self.p_mlp[i], self._drop_seed, self._drop_offset(3 + 4 * i), out=self.branch, keep=self.keep_b[i])
        # final norm on the CLS rows only (vit.py:416-419) with the last residual add fused, fp32 head + mean CE
        ops.layernorm_fwd(self.res, self.gf, self.bf_, self.eps, out=self.xn, mean=self.meanf, rstd=self.rstdf,
                          rows=B, row_stride=N * D, add=self.branch_cls if self.cls_only_tail else self.branch,
                          add_row_stride=D if self.cls_only_tail else None, x_out=self.res, add_scale=dps(2 * self.L - 1), scale_period=1,
                          drop=dsite(3 + 4 * (self.L - 1), self.p_mlp[-1]) + (N * D,) if fused and self.p_mlp[-1] > 0 and not self.swiglu else None)
        ops.sgemm_small(self.xn, self._param_view("fc.weight"), trans_b=True, bias=self._param_view("fc.bias"),
                        out=self.logits)
        ops.cross_entropy(self.logits, self.targets if self.soft_targets else self.labels, dlogits=self.dlogits,
                          row_loss=self.row_loss, loss=self.loss)
        if self.dynamic_scale:
            self.dlogits.mul_(self.scaler[6])   # device scalar written by the previous optimizer step (graph-safe)
        elif self.loss_scale != 1.0:
            self.dlogits.mul_(self.loss_scale)  # every gradient carries the scale until the optimizer divides it out

    def _gemm_rows(self, a, w, bias=None, **kw):
        """GEMM on the B CLS rows of the last block: the split-K few-row kernel where it applies (STORE / GELU / MUL epilogues,
        K % 128 == 0), else apla_gemm_nt."""
        epi = kw.get("epilogue", ops.EPI_STORE)
        M, K, N = a.shape[0], a.shape[1], w.shape[0]
        if epi in (ops.EPI_STORE, ops.EPI_GELU, ops.EPI_MUL) and self.cls_small_gemm:
            key = (M, N, K)
            if key not in self._small_ws:
                self._small_ws[key] = ops.gemm_small_workspace(M, N, K, self.device)
            ws = self._small_ws[key]
            if ws is not None:
                return ops.gemm_nt_small(a, w, bias, workspace=ws, **kw)
        return ops.gemm_nt(a, w, bias, **kw)

    def _forward_last_block_tail(self, st, i, dps=lambda k: None):
        """Block L-1 after its qkv GEMM.  The head reads only x[:, 0] of this block's output (vit.py:416-419) and rows do not
        mix after the attention, so everything from the attention on runs for the B CLS rows only: one query per head
        against all keys (apla_attn_fwd_cls), then projection, LN2, MLP on [B, D] instead of [B*N, D] (about 0.28 TFLOP
        less per step at cfg 2).  The saved activations land where the (already CLS-only) backward of this block reads them:
        rows b*N of o / xmid / act_saved, lse[:, :, 0], compact LN2 statistics."""
        B, N, H, D = self.B, self.N, self.H, self.D
        cls = lambda t: t.view(B, -1)[:, :t.shape[1]]          # rows b*N of a [B*N, C] buffer as a strided [B, C] view
        ops.attn_fwd_cls(self.qkv[i], B, N, H, self.scale, o=self.o[i], lse=self.lse[i])
        self._gemm_rows(cls(self.o[i]), st.Wnat, st.bnat, out=self.branch_cls)
        ops.layernorm_fwd(self.res, None, None, self.eps, out=self.ln_cls, mean=self.mean2_cls, rstd=self.rstd2_cls,
                          rows=B, row_stride=N * D, add=self.branch_cls, add_row_stride=D, x_out=self.res, D=D,
                          add_scale=dps(2 * i), scale_period=1)
        if self.swiglu:
            ops.gemm_nt(self.ln_cls, st.W12, st.b12, epilogue=ops.EPI_SWIGLU, aux_out=cls(self.act_saved[i]), out=self.h_cls)
        else:
            self._gemm_rows(self.ln_cls, st.Wfc1, st.bfc1, epilogue=ops.EPI_GELU, aux_out=cls(self.act_saved[i]), out=self.h_cls)
        self._gemm_rows(self.h_cls, st.Wout, st.bout, out=self.branch_cls)

    def _backward_head(self):
        B, N, D = self.B, self.N, self.D
        ops.sgemm_small(self.dlogits, self.xn, trans_a=True, out=self._grad_view("fc.weight"))
        ops.colsum(self.dlogits, out=self._grad_view("fc.bias"))
        ops.sgemm_small(self.dlogits, self._param_view("fc.weight"), out=self.dxn)
        # only the CLS rows of the residual gradient exist below the final norm: they are written here, read by the CLS-only
        # backward of the last block, and the first dense LayerNorm backward takes the other rows as zero (dres_period = N) —
        # the gradient stream is never zero-filled
        ops.layernorm_bwd(self.dxn, self.res, self.gf, self.meanf, self.rstdf, out=self.G,
                          out_bf16=None if self.Gb is self.G else self.Gb, rows=B, row_stride=N * D)

    def _proj_dw(self, st, dyg, o):
        """dW1 / db1 of one block into the flat gradient buffer (through a padded temporary when r is not a multiple of 64)."""
        if st.r_pad == st.r:
            ops.proj_dw(dyg, o, self._grad_view(st.W1_name), self._grad_view(st.b1_name), row_scale=st.row_scale, workspace=self.dw_ws)
            return
        dW, db = self.dW_tmp[:st.r_pad], self.db_tmp[:st.r_pad]
        ops.proj_dw(dyg, o, dW, db, row_scale=st.row_scale, workspace=self.dw_ws)
        self._grad_view(st.W1_name).copy_(dW[:st.r])
        self._grad_view(st.b1_name).copy_(db[:st.r])

    def _flush_dw(self):
        """The pending blocks' dW1 / db1 in one launch pair (end of a backward segment, or DW_BATCH blocks collected)."""
        pend, self._dw_pending = self._dw_pending, []
        if not pend:
            return
        if len(pend) == 1:
            return self._proj_dw(*pend[0])
        nb, st0 = len(pend), pend[0][0]
        if nb not in self.dw_ws_batched:
            self.dw_ws_batched[nb] = ops.dw_workspace_batched(self.M, st0.r_pad, self.D, nb, self.device)
        ops.proj_dw_batched([p[1] for p in pend], [p[2] for p in pend], [self._grad_view(p[0].W1_name) for p in pend],
                            [self._grad_view(p[0].b1_name) for p in pend], row_scale=[p[0].row_scale for p in pend],
                            workspace=self.dw_ws_batched[nb])

    def _backward_block(self, i):
        st = self.blocks[i]
        B, N, H, M, D = self.B, self.N, self.H, self.M, self.D
        copy = None if self.Gb is self.G else self.Gb
        mlp_drop = self.drop_on and self.p_mlp[i] > 0 and not self.swiglu
        proj_drop = self.drop_on and self.p_proj[i] > 0
        fused = self.drop_fused
        dsite = lambda site, p_: (self.drop_rng, self._drop_stride, site, p_)       # noqa: E731
        # gradient entering the MLP branch: through the mask of the dropout after fc2 (the one after the activation lives in the saved GELU').
        # Fused form: the LayerNorm backward (or, for the last block, the pass in _segment) that produced the stream wrote the masked copy.
        if mlp_drop and not fused:
            g_in = ops.dropout_bwd(self.Gb, self.keep_b[i], self.p_mlp[i], out=self.Gm)
        else:
            g_in = self.Gm if mlp_drop else self.Gb
        if self.swiglu:
            ops.gemm_nt(g_in, st.WoutT, None, epilogue=ops.EPI_SWIGLU_BWD, aux_in=self.act_saved[i], out=self.dact_out)
            ops.gemm_nt(self.dact_out, st.Wdfc1_i, None, out=self.dln, tag=ops.TAG_DFC1)
        else:
            ops.gemm_nt(g_in, st.WoutT, None, epilogue=ops.EPI_MUL, aux_in=self.act_io[i], out=self.dact_out)
            ops.gemm_nt(self.dact_out, st.Wdfc1_i, None, out=self.dln, tag=ops.TAG_DFC1)
        dyg = (self.dyg_all[i] if self.dw_batch > 1 else self.dyg)[:M * st.r_pad].view(M, st.r_pad)
        dps = (lambda k: self.dp_scale[k]) if self.dp_on else (lambda k: None)      # stochastic depth: the branches' per-sample factors
        if proj_drop and fused:
            # one kernel: the stream, its copy through proj_drop's mask x the attention branch's DropPath factor (the dX operand), and the
            # gathered columns of dW1 taken from that copy
            ops.layernorm_bwd(self.dln, self.xh2[i], None, None, self.rstd2[i], dres=self.G, out=self.G, inds=st.inds, r=st.r_pad, gathered=dyg,
                              dy_scale=dps(2 * i + 1), scale_period=N, masked=self.Gm, mask_scale=dps(2 * i), drop=dsite(1 + 4 * i, self.p_proj[i]))
        elif proj_drop:
            # the projection's output gradient passes proj_drop's mask (and the attention branch's DropPath factor) BEFORE both of its
            # consumers — the gathered columns of dW1 and the dX product —, so the LayerNorm backward gathers nothing here
            ops.layernorm_bwd(self.dln, self.xh2[i], None, None, self.rstd2[i], dres=self.G, out=self.G, out_bf16=copy,
                              dy_scale=dps(2 * i + 1), scale_period=N)
            ops.dropout_bwd(self.Gb, self.keep_a[i], self.p_proj[i], out=self.Gm)
            if self.dp_on:
                ops.scale_samples(self.Gm.view(B, N * D), self.dp_scale[2 * i], out=self.Gm.view(B, N * D))
            ops.gather_cols(self.Gm, st.inds, st.r_pad, out=dyg)
        else:
            ops.layernorm_bwd(self.dln, self.xh2[i], None, None, self.rstd2[i], dres=self.G, out=self.G,
                              out_bf16=copy, inds=st.inds, r=st.r_pad, gathered=dyg, dy_scale=dps(2 * i + 1), gather_scale=dps(2 * i), scale_period=N)
        if self.dw_batch > 1:
            self._dw_pending.append((st, dyg, self.o[i]))
            if len(self._dw_pending) == self.dw_batch:
                self._flush_dw()
        else:
            self._proj_dw(st, dyg, self.o[i])
        if i == 0:
            return  # nothing upstream of block 0's projection is trainable
        ops.gemm_nt(self.Gm if proj_drop else self.Gb, st.WnatT_i, None, out=self.dO, tag=ops.TAG_DPROJ)
        if self.drop_on and self.p_attn[i] > 0:
            ops.attn_bwd_dropout(self.qkv[i], self.o[i], self.dO, self.lse[i], B, N, H, self.scale, self.p_attn[i], self._drop_seed,
                                 self._drop_offset(4 + 4 * i), dqkv=self.dqkv, delta=self.delta)
        else:
            ops.attn_bwd(self.qkv[i], self.o[i], self.dO, self.lse[i], B, N, H, self.scale, dqkv=self.dqkv, delta=self.delta)
        ops.gemm_nt(self.dqkv, st.WqkvT_i, None, out=self.dln, tag=ops.TAG_DQKV)
        nxt = fused and i > 0 and self.p_mlp[i - 1] > 0 and not self.swiglu     # the stream's next consumer is block i - 1's MLP branch
        ops.layernorm_bwd(self.dln, self.xh1[i], None, None, self.rstd1[i], dres=self.G, out=self.G, out_bf16=copy,
                          dy_scale=None if proj_drop else dps(2 * i), scale_period=N,    # (with proj_drop the factor is already in Gm)
                          masked=self.Gm if nxt else None, drop=dsite(3 + 4 * (i - 1), self.p_mlp[i - 1]) if nxt else None)

    def _backward_last_block(self):
        """Backward of block L-1 exploiting that only the CLS rows (token 0 of every sequence) of the incoming residual
        gradient are non-zero (final norm + x[:,0], vit.py:416-419): the MLP backward, LN2 backward, dW1 and the projection
        dX run on B rows instead of B*N, and the attention backward is rank-1 per head (apla_attn_bwd_cls).  From dqkv on
        the gradient is dense again."""
        i = self.L - 1
        st = self.blocks[i]
        B, N, H, D = self.B, self.N, self.H, self.D
        cls = lambda t: t.view(B, -1)[:, :t.shape[1]]          # rows b*N of a [B*N, C] buffer as a strided [B, C] view
        copy = None if self.Gb is self.G else self.Gb
        if self.swiglu:
            ops.gemm_nt(cls(self.Gb), st.WoutT, None, epilogue=ops.EPI_SWIGLU_BWD, aux_in=cls(self.act_saved[i]), out=self.dact_cls)
            ops.gemm_nt(self.dact_cls, st.W12T, None, out=self.dln_cls)
        else:
            self._gemm_rows(cls(self.Gb), st.WoutT, None, epilogue=ops.EPI_MUL, aux_in=cls(self.act_saved[i]), out=self.dact_cls)
            self._gemm_rows(self.dact_cls, st.Wfc1T, None, out=self.dln_cls)
        dyg = self.dyg_cls[:B * st.r_pad].view(B, st.r_pad)
        if self.cls_only_tail:    # the forward normalised the CLS rows only: compact [B, D] xhat
            xh, xh_stride, rstd = self.ln_cls, D, self.rstd2_cls
        else:                     # (diagnostic full forward) rows b*N of the dense buffers
            xh, xh_stride, rstd = self.xh2[i], N * D, self.rstd2[i][::N].contiguous()
        dps = (lambda k: self.dp_scale[k]) if self.dp_on else (lambda k: None)
        ops.layernorm_bwd(self.dln_cls, xh, None, None, rstd, dres=self.G, out=self.G, out_bf16=copy, inds=st.inds, r=st.r_pad,
                          gathered=dyg, rows=B, row_stride=N * D, x_row_stride=xh_stride,
                          dy_scale=dps(2 * i + 1), gather_scale=dps(2 * i), scale_period=1)
        self._proj_dw(st, dyg, cls(self.o[i]))
        if i == 0:
            return
        self._gemm_rows(cls(self.Gb), st.WnatT, None, out=self.dO_cls)
        ops.attn_bwd_cls(self.qkv[i], self.o[i], self.dO_cls, self.lse[i], B, N, H, self.scale, dqkv=self.dqkv)
        ops.gemm_nt(self.dqkv, st.WqkvT_i, None, out=self.dln, tag=ops.TAG_DQKV)
        # first dense backward of the step: the incoming stream holds the CLS rows only (see _backward_head)
        ops.layernorm_bwd(self.dln, self.xh1[i], None, None, self.rstd1[i], dres=self.G, out=self.G, out_bf16=copy, dres_period=N,
                          dy_scale=dps(2 * i), scale_period=N)

    def _segment(self, k):
        """Segment k of the step: segment 0 = weight re-scatter + forward + head + the backward down to block seg_cuts[0];
        segment k > 0 = the backward of blocks seg_cuts[k-1]-1 .. seg_cuts[k]."""
        with ops.reserved_cus(self.reserve_cus):
            if k == 0:
                self.refresh_weights()
                self._forward()
                if self.drop_on:    # every block runs its dense backward: the stream below the final norm is zero outside the CLS rows
                    self.G.zero_()
                    if self.Gb is not self.G:
                        self.Gb.zero_()
                    self._backward_head()
                    if self.drop_fused and self.p_mlp[-1] > 0 and not self.swiglu:    # the last block's MLP branch: no LayerNorm backward above it writes the masked copy
                        ops.dropout_dev(self.Gb, self.p_mlp[-1], self.drop_rng, self._drop_stride, 3 + 4 * (self.L - 1), out=self.Gm)
                    self._backward_block(self.L - 1)
                else:
                    self._backward_head()
                    self._backward_last_block()
                hi = self.L - 1
            else:
                hi = self.seg_cuts[k - 1]
            for i in range(hi - 1, self.seg_cuts[k] - 1, -1):
                self._backward_block(i)
            if self.dw_batch > 1:
                self._flush_dw()   # a segment's gradients are complete when it ends (its all-reduce chunk is launched next)

    def _capture(self):
        n = len(self.seg_cuts)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture (lazy module loading, allocator)
            for k in range(n):
                self._segment(k)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs = []
        for k in range(n):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **({"pool": graphs[0].pool()} if graphs else {})):
                self._segment(k)
            graphs.append(g)
        self._graphs = tuple(graphs)

    # ------------------------------------------------------------------ public API
    def set_batch(self, images: torch.Tensor, labels: torch.Tensor):
        self.images.copy_(images, non_blocking=True)
        if self.soft_targets:
            if labels.ndim != 2:
                raise ValueError("this engine was built with soft_targets=True: pass probability targets [B, C]")
            self.targets.copy_(labels.float(), non_blocking=True)
        else:
            if labels.ndim != 1:
                raise ValueError("class-id labels [B] expected (build the engine with soft_targets=True for probability targets)")
            self.labels.copy_(labels.to(torch.int32), non_blocking=True)

    @property
    def _drop_stride(self) -> int:
        return 4 * self.L + 8

    def _sync_drop_rng(self):
        to_i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v      # noqa: E731  (the kernels read the words as unsigned)
        self.drop_rng.copy_(torch.tensor([to_i64(self._drop_seed & ((1 << 64) - 1)), self._drop_step], dtype=torch.int64))

    def _drop_offset(self, site: int) -> int:
        """Philox offset of a dropout site in the current step: sites 0 = pos_drop, 1 + 4 i / 2 + 4 i / 3 + 4 i / 4 + 4 i = block i's
        proj_drop / Mlp.drop after the activation / Mlp.drop after fc2 / attn_drop; a fresh set per step (`_drop_step` counts from 1)."""
        return self._drop_step * self._drop_stride + site

    def set_dropout_seed(self, seed: int, step: int = 0):
        """Tests / reproducibility: the key of every dropout mask (and the step counter the offsets derive from)."""
        self._drop_seed, self._drop_step = int(seed), int(step)
        self._sync_drop_rng()

    def set_drop_path_uniforms(self, u: Optional[torch.Tensor]):
        """Tests / reproducibility: the uniform numbers u [2 L, B] of the NEXT steps' DropPath draws (row 2 i / 2 i + 1 = block i's
        attention / MLP branch; vit.py:74-82: factor = floor(keep_prob + u) / keep_prob) instead of the engine's own generator;
        None returns to the generator."""
        self._dp_given = None if u is None else u.to(self.device, torch.float32).reshape(2 * self.L, self.B)

    def _draw_drop_path(self):
        """One factor per (block, branch, sample) for this step, written into the static buffer the (captured) LayerNorm launches read."""
        u = self._dp_given if self._dp_given is not None else torch.rand(2 * self.L, self.B, device=self.device, generator=self._dp_gen)
        torch.div(torch.floor(self._dp_keep + u), self._dp_keep, out=self.dp_scale)

    @_half_mode
    def forward_backward(self):
        """Forward + loss + backward of the batch in self.images/self.labels; grads land in the flat buffer.
        With world > 1 the gradient all-reduce of each chunk is launched on a side stream as soon as its segment of the
        backward has been enqueued."""
        if self.dp_on:
            self._draw_drop_path()
        if self.drop_on:
            self._drop_step += 1
            self.drop_rng[1:].add_(1)       # outside the captured launches: they read {seed, step} from this buffer
        if self.use_graphs and self._graphs is None:
            self._capture()
        for k in range(len(self.seg_cuts)):
            if self.use_graphs:
                self._graphs[k].replay()
            else:
                self._segment(k)
            self.exchanger.launch_chunk(k)   # RCCL all-reduce of what this segment produced, on the side stream
        self.exchanger.wait()

    @_half_mode
    def optimizer_step(self, lr: Optional[float] = None):
        """Fused clip + AdamW on the flat buffer (DDP mean = grad_scale 1/world)."""
        self.step_count += 1
        oc = self.optim
        if self.dynamic_scale:
            ops.adamw_step_dynamic(self.flat_params, self.flat_grads, self.exp_avg, self.exp_avg_sq, self.decay_mask,
                                   self.scaler, self._scaler_calls & 1, lr=oc.lr if lr is None else lr,
                                   weight_decay=oc.weight_decay, betas=oc.betas, eps=oc.eps,
                                   max_norm=oc.grad_clipping or 0.0, grad_scale=1.0 / self.world, norm_ws=self.norm_ws)
            self._scaler_calls += 1
            return
        ops.adamw_step(self.flat_params, self.flat_grads, self.exp_avg, self.exp_avg_sq, self.decay_mask,
                       lr=oc.lr if lr is None else lr, weight_decay=oc.weight_decay, betas=oc.betas, eps=oc.eps,
                       step=self.step_count, max_norm=oc.grad_clipping or 0.0, grad_scale=1.0 / (self.world * self.loss_scale),
                       norm_ws=self.norm_ws)

    @_half_mode
    def train_step(self, images=None, labels=None, lr: Optional[float] = None):
        if images is not None:
            self.set_batch(images, labels)
        self.forward_backward()
        self.optimizer_step(lr)
        return self.loss

    @_half_mode
    def forward_only(self, images, labels=None):
        """Inference on one batch (the reference's ``model(images, return_embedding=True)`` under no_grad,
        defaults/trainer.py:207-210, models.py:81-92): the step's own forward launch sequence with the current trainable rows,
        no backward.  Returns (logits [B, C] fp32, features [B, D] fp32 = the final-norm CLS token, mean CE loss or None);
        the tensors are the engine's buffers — clone what must outlive the next call."""
        b = images.shape[0]
        if b > self.B or b < 1:
            raise ValueError(f"this engine was built for batches of at most {self.B} images (got {b})")
        self.images[:b].copy_(images, non_blocking=True)      # a short (last) batch: the other rows keep the previous batch — samples do not mix
        if labels is not None and not self.soft_targets:
            self.labels[:b].copy_(labels.to(torch.int32), non_blocking=True)
        self.refresh_weights()
        self._forward(inference=True)
        if b == self.B:
            return self.logits, self.xn, (self.loss if labels is not None and not self.soft_targets else None)
        loss = None
        if labels is not None and not self.soft_targets:     # the kernel's loss is the mean over all B rows: redo it over the b real ones
            loss = torch.nn.functional.cross_entropy(self.logits[:b], labels.to(self.device).long()).reshape(1)
        return self.logits[:b], self.xn[:b], loss

    @_half_mode
    def time_fc1_launches(self, steps: int = 3) -> float:
        """Average duration (ms) of the fc1 GEMM + activation launch — the largest single kernel of the step — measured with
        HIP events recorded on the launch stream around every such launch of `steps` further training steps (launched
        eagerly: events cannot be read back from a replayed graph).  This is the in-context figure a rocprofv3 kernel trace of
        the same run reports for this kernel; a back-to-back microbenchmark of the same launch runs ~8 % faster."""
        self._fc1_events = []
        try:
            for _ in range(steps):
                for k in range(len(self.seg_cuts)):
                    self._segment(k)
                    self.exchanger.launch_chunk(k)
                self.exchanger.wait()
                self.optimizer_step()
            torch.cuda.synchronize(self.device)
            ts = [a.elapsed_time(b) for a, b in self._fc1_events]
        finally:
            self._fc1_events = None
        return sum(ts) / max(len(ts), 1)

    @property
    def grad_norm(self):
        return self.norm_ws[1]

    @property
    def skipped_steps(self) -> int:
        """Optimizer steps skipped so far because the gradient norm was not finite (static loss-scale / bf16 path; the dynamic
        scaler keeps its own count).  Reads device memory: call it at logging cadence, not every step."""
        if self.dynamic_scale:
            # calls since the last (re)start minus the steps the scaler counted since then (a resumed session starts its step
            # slot at the restored count: checkpoint.load_session)
            taken = float(self.scaler[3 * (self._scaler_calls & 1) + 2]) - self._scaler_step0
            return int(self._scaler_calls - taken) if self._scaler_calls else 0
        return int(float(self.norm_ws[260 + ((self.step_count + 1) & 1)]))

    @property
    def applied_steps(self) -> int:
        """Updates actually applied = what torch.optim.AdamW would report as ``step`` (skipped steps do not count)."""
        if self.dynamic_scale:   # the scaler's own count (includes the steps of a resumed session)
            return int(float(self.scaler[3 * (self._scaler_calls & 1) + 2]))
        return self.step_count - self.skipped_steps

    def grads(self):
        return {n: self._grad_view(n) for n in self.names}

    def peak_memory_bytes(self):
        return torch.cuda.max_memory_allocated(self.device)
