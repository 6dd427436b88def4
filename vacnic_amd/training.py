"""One VACNIC training step on gfx950 kernels — the body of train_epoch (TRAIN:253-383) without the
host syncs: ViT features -> multimodal BART (+ fused lm_head/CE) -> frozen guide BART -> CoLaM ->
SECLA -> backward -> DDP all-reduce -> fused AdamW with on-device linear-warmup schedule.
"""
from dataclasses import dataclass

import torch

from . import kernels as K
from . import ops
from . import streams
from .config import ClipVisionConfig, VacnicConfig
from .ddp import DistributedDataParallel
from .models.clip_vit import CLIPVisualOnly, extract_clip_img_feat
from .models.guide_bart import BartForConditionalGeneration
from .models.mmbart import BartForMultiModalGeneration


@dataclass
class TrainArgs:
    """the trainer flags that shape the step (names/defaults of TRAIN:5-82, values of run_full_train.sh)."""
    lr_bart: float = 3e-5
    weight_decay: float = 0.01
    warmup_rate: float = 0.05
    num_training_steps: int = 100000
    margin: float = 1.0
    alpha: float = 0.5
    mapping_loss_weight: float = 1.0
    use_secla: bool = True
    no_mapping: bool = False
    no_clip_norm: bool = True
    clip_norm: float = 0.1
    prompt_mlp_type: str = "clipcap"


class FusedAdamW:
    """optim.AdamW(betas=(0.9,0.999), eps=1e-8) + get_linear_schedule_with_warmup (TRAIN:91,99-107) over the
    gradient arena: one lr kernel + one AdamW kernel per step, lr and step counter in device memory."""

    def __init__(self, arena, lr, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8, num_warmup_steps=0.0,
                 num_training_steps=1.0, world_size=1):
        self.arena = arena
        arena.init_optimizer_state()
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.warmup, self.total = float(num_warmup_steps), float(num_training_steps)
        self.world = world_size
        self.hyper = torch.zeros(2, device=arena.device, dtype=torch.float32)      # {lr, step}
        self.clip = None                  # {clip coefficient, total grad norm} of the last clipped step (device)
        self._clip_scratch = None

    def step(self, clip_norm=None):
        """clip_norm: max total gradient norm (torch.nn.utils.clip_grad_norm_, TRAIN:365-366) or None.  The norm is taken
        over the averaged gradient (after the 1/world scaling), as DDP hands it to clip_grad_norm_ in the reference; the
        coefficient stays in device memory and is applied where the AdamW kernel reads the gradient."""
        a = self.arena
        K.lr_step(self.hyper, self.lr, self.warmup, self.total, ops.Rng.device_counter())   # also advances the dropout counter
        clip = None
        if clip_norm is not None:
            if self.clip is None:
                self.clip = torch.zeros(2, device=a.device, dtype=torch.float32)
                self._clip_scratch = torch.empty(1024, device=a.device, dtype=torch.float32)
            clip = K.grad_clip_coef(a.grad, a.n, float(clip_norm), 1.0 / self.world, self._clip_scratch, self.clip)
        K.adamw(a.flat32, a.grad, a.exp_avg, a.exp_avg_sq, a.flat16, self.hyper, a.n, self.betas[0], self.betas[1],
                self.eps, self.wd, grad_scale=1.0 / self.world, zero_grad=True, clip_coef=clip)

    def begin_step(self):
        """first half of step() for a range-wise update (DistributedDataParallel.reduce_and_step): schedule + counters once."""
        K.lr_step(self.hyper, self.lr, self.warmup, self.total, ops.Rng.device_counter())

    def step_range(self, start, end):
        """AdamW on arena elements [start, end) (a DDP bucket whose all-reduce has finished); begin_step() comes first."""
        a = self.arena
        K.adamw(a.flat32[start:end], a.grad[start:end], a.exp_avg[start:end], a.exp_avg_sq[start:end], a.flat16[start:end], self.hyper,
                end - start, self.betas[0], self.betas[1], self.eps, self.wd, grad_scale=1.0 / self.world, zero_grad=True)

    def zero_grad(self):
        pass            # fused into step(): the AdamW kernel clears the gradient arena it just consumed

    def state_dict(self):
        return {"exp_avg": self.arena.exp_avg, "exp_avg_sq": self.arena.exp_avg_sq, "hyper": self.hyper}


def build_models(cfg: VacnicConfig, vcfg: ClipVisionConfig, device="cuda", seed=0, init="device", state_dicts=None, with_guide=True):
    """Random-init (or state-dict) construction of the three networks of the step, finalized in HBM arenas.
    init='device': N(0, 0.02) drawn on the GPU (fast, for benchmarks); init='synthetic': name-keyed numpy
    weights identical to the oracle's (parity tests)."""
    from . import synthetic
    clip_model = CLIPVisualOnly(vcfg)
    model = BartForMultiModalGeneration(cfg, enc_fusion_layer=cfg.enc_fusion_layer, dim_common=cfg.dim_common, img_size=768,
                                        prompt_mlp_type=cfg.prompt_mlp_type, map_size=cfg.map_size, prompt_size=cfg.prompt_size, clip_model=None,
                                        freeze_clip=True, max_ner_type_len=cfg.max_ner_type_len,
                                        max_ner_type_len_gt=cfg.max_ner_type_len_gt, only_image=cfg.only_image,
                                        init_attn_weight=cfg.init_attn_weight)
    guide = BartForConditionalGeneration(cfg) if with_guide else None      # with_guide=False: inference (no CoLaM teacher)
    if init == "synthetic" or state_dicts is not None:
        sds = state_dicts or (synthetic.make_state_dict(synthetic.mmbart_param_shapes(cfg), seed=seed + 1),
                              synthetic.make_state_dict(synthetic.guide_bart_param_shapes(cfg), seed=seed + 2),
                              synthetic.make_state_dict(synthetic.clip_visual_param_shapes(vcfg), seed=seed + 4, std=0.05))
        load_named(model, sds[0]); load_named(clip_model.visual, sds[2])
        if guide is not None:
            load_named(guide, sds[1])
    model.finalize(device)
    if guide is not None:
        guide.finalize(device)
    clip_model.finalize(device)
    if init == "device" and state_dicts is None:
        # one generator per network: the frozen CLIP tower a trainer drew from `seed` is reproducible without building the guide
        for k, (m, std) in enumerate(((model, cfg.init_std), (guide, cfg.init_std), (clip_model.visual, 0.02))):
            if m is None:
                continue
            g = torch.Generator(device=device).manual_seed(seed + 7919 * k)
            for name, p in m.named_parameters():
                if p.dim() >= 2 or "embedding" in name:
                    p.data.normal_(0.0, std, generator=g)
                elif name.endswith("weight"):
                    p.data.fill_(1.0)
                else:
                    p.data.zero_()
            m.arena.refresh_shadow()
    model.clip_model = clip_model                     # the reference keeps CLIP inside the model (MFULL:1892; TRAIN:274)
    return model, guide, clip_model


def load_named(module, sd):
    params = dict(module.named_parameters())
    missing = [k for k in params if k not in sd and not k.endswith("embed_tokens.weight") and k != "lm_head.weight"]
    if missing:
        raise KeyError(f"state dict lacks {missing[:5]} ...")
    with torch.no_grad():
        for k, p in params.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(p.shape):
                    raise ValueError(f"shape mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(p.shape)}")
                p.data.copy_(sd[k])


def to_device(batch, device):
    return {k: v.to(device, non_blocking=True) for k, v in batch.items()}


def image_feature_index(cfg):
    """which output of extract_clip_img_feat feeds `image_features`: the ln_post CLS vector for the ClipCap prompt MLP, the
    ln_post patch tokens for `--prompt_mlp_type mlp` (the reference trainers branch the same way at every model call)."""
    return 1 if cfg.prompt_mlp_type == "clipcap" else 0


_PLAN = None          # the PlannedTrainStep being recorded (forward_losses leaves marks in it)


def forward_losses(model, guide, batch, args: TrainArgs, ready=None, towers=None):
    """Forward of one step; returns (total, out4={total, txt, secla, colam}, model_out).  `model` may be the DDP wrapper
    (like TRAIN:274 `model.module`).  `ready`: optional event after which the batch tensors are valid in HBM; with side
    streams enabled the frozen towers then start on it instead of on the compute stream's tail (= the previous AdamW)."""
    net = model.module if isinstance(model, DistributedDataParallel) else model
    cfg = net.config
    feat = image_feature_index(cfg)
    if not args.no_mapping and not args.use_secla and not cfg.only_image:
        # TRAIN:331-345 (`--use_secla False`): re-runs the model with add_ner_ffn=False, which the reference's encoder rejects
        # with a mask-shape ValueError (see BartEncoderLayer) — there is no working behaviour to reproduce
        raise ValueError("--use_secla False with --no_mapping False: the reference's pooled face-name branch (TRAIN:331-345) "
                         "fails inside its own encoder (add_ner_ffn=False, attention-mask size); use --use_secla True or --no_mapping True")
    src, tgt = batch["article_ids"], batch["caption_ids"]
    main = torch.cuda.current_stream()
    aux, vis = streams.aux_stream(), streams.vit_stream()
    if aux is None:
        if ready is not None:
            main.wait_event(ready)           # single-stream schedule: the loader's copy stream still has to hand the batch over
        src_mask, _ = K.prep_ids(src, cfg.pad_token_id)                                      # create_src_mask_bart, TRAIN:268
        tgt_mask, tgt_in = K.prep_ids(tgt, cfg.pad_token_id, start_id=cfg.eos_token_id)     # shift_tokens_right, TRAIN:267,296
        img_cls = extract_clip_img_feat(net.clip_model, batch["img_tensor"])[feat]            # TRAIN:274-276
    elif towers is not None and _PLAN is not None:
        # recording a launch plan around the tower graphs (PlannedTrainStep): the graphs are replayed by the host before each plan
        # replay and joined at the plan's marks; the id preprocessing runs inside the plan, on the compute stream
        src_mask, _ = K.prep_ids(src, cfg.pad_token_id)
        tgt_mask, tgt_in = K.prep_ids(tgt, cfg.pad_token_id, start_id=cfg.eos_token_id)
        img_cls, gh = towers.img_cls, towers.gh
    elif towers is not None:
        # frozen towers as two hipGraph replays on their side streams (FrozenTowerGraphs)
        src_mask, tgt_mask, tgt_in, ev_prep = towers.launch(batch, ready)
        main.wait_event(ev_prep)
        main.wait_event(towers.ev_vit)
        img_cls, gh = towers.img_cls, towers.gh
        for tns in (src_mask, tgt_mask, tgt_in):
            tns.record_stream(main)
    elif streams.explicit():
        # explicit scheduling (see streams.explicit): the same schedule as the branch below — id preprocessing + frozen guide
        # forward on the aux stream, frozen ViT on its own stream — with kernels.launch_on / kernels.fence instead of
        # torch.cuda.stream contexts and Tensor.record_stream, so that the whole step can be recorded into a launch plan
        main_raw, aux_raw, vis_raw = K._stream(), streams.raw("aux"), streams.raw("vit")
        if ready is None:
            K.fence(main_raw, aux_raw)
            K.fence(main_raw, vis_raw)
        else:                                   # (a torch event of the loader: not part of a plan — plans refresh their inputs in place)
            aux.wait_event(ready)
            vis.wait_event(ready)
        with K.launch_on(aux_raw):
            src_mask, _ = K.prep_ids(src, cfg.pad_token_id)
            tgt_mask, tgt_in = K.prep_ids(tgt, cfg.pad_token_id, start_id=cfg.eos_token_id)
        K.fence(aux_raw, main_raw)              # the student needs the masks now, the guide's output only at the CoLaM loss
        with K.launch_on(vis_raw):              # the student's encoder input needs the image feature: ViT goes first
            img_cls = extract_clip_img_feat(net.clip_model, batch["img_tensor"])[feat]
        K.fence(vis_raw, main_raw)              # (before the guide is enqueued: the towers may share one stream)
        if guide is not None:
            with K.launch_on(aux_raw):
                gh = guide(input_ids=src, attention_mask=src_mask, decoder_input_ids=tgt_in)["decoder_hidden_states"][-1]   # TRAIN:293-294
    else:
        # id preprocessing + frozen guide forward on the aux stream, frozen ViT on its own stream: they depend only on
        # the batch, so they fill the bubbles of the main chain (and of the previous step's AdamW)
        for s_ in (aux, vis):
            if ready is None:
                s_.wait_stream(main)
            else:
                s_.wait_event(ready)
        with torch.cuda.stream(aux):
            src_mask, _ = K.prep_ids(src, cfg.pad_token_id)
            tgt_mask, tgt_in = K.prep_ids(tgt, cfg.pad_token_id, start_id=cfg.eos_token_id)
            ev_prep = torch.cuda.Event()
            ev_prep.record(aux)
            if guide is not None:
                gh = guide(input_ids=src, attention_mask=src_mask, decoder_input_ids=tgt_in)["decoder_hidden_states"][-1]   # TRAIN:293-294
        with torch.cuda.stream(vis):
            img_cls = extract_clip_img_feat(net.clip_model, batch["img_tensor"])[feat]
        main.wait_event(ev_prep)
        main.wait_stream(vis)
        for tns in (src_mask, tgt_mask, tgt_in, img_cls):
            tns.record_stream(main)
        for tns in (src, tgt):
            tns.record_stream(aux)
        batch["img_tensor"].record_stream(vis)
    kw = {}
    if not cfg.only_image:
        names_mask, _ = K.prep_ids(batch["names_art_ids"], cfg.pad_token_id)                 # TRAIN:270
        kw = dict(face_features=batch["face_emb"], face_mask=K.face_mask(batch["face_emb"]), name_ids=batch["names_art_ids"],
                  name_mask=names_mask)
    out = model(input_ids=src, attention_mask=src_mask, decoder_input_ids=tgt_in, image_features=img_cls, labels=tgt,
                output_logits=False, add_ner_ffn=True, **kw)                                  # TRAIN:281 + fused CE (TRAIN:287)
    txt = out["loss"]
    colam = secla = None
    if guide is not None:
        if towers is not None and _PLAN is not None:
            _PLAN.mark("join_guide")             # the host makes the compute stream wait for the guide graph here
        elif aux is not None and towers is None and streams.explicit():
            K.fence(streams.raw("aux"), K._stream())
        elif aux is not None:
            main.wait_stream(aux)
            if towers is None:
                gh.record_stream(main)
        else:
            gh = guide(input_ids=src, attention_mask=src_mask, decoder_input_ids=tgt_in)["decoder_hidden_states"][-1]   # TRAIN:293-294
        colam = ops.ColamFn.apply(out["decoder_hidden_states"][-1], gh, tgt_mask, args.margin, args.alpha)        # TRAIN:296-307
    if towers is not None and _PLAN is not None:
        _PLAN.mark("towers_consumed")
    elif towers is not None:
        towers.mark_consumed()               # static img_cls / gh have been read: the next replay may overwrite them
    if args.use_secla and not args.no_mapping and not cfg.only_image:
        enc = net.model.encoder
        ln = enc.layernorm_embedding_ner
        names = K.name_embed_mean(batch["names_ids"], enc.embed_tokens_ner.weight.w16, enc.embed_positions_ner.weight.w16,
                                  ln.weight.data, ln.bias.data, embed_scale=enc.embed_scale)   # get_embedding_ner, TRAIN:327
        secla = ops.SeclaFn.apply(out["hidden_states_face"], names, args.mapping_loss_weight)  # TRAIN:329
    total, out4 = ops.total_loss(txt, secla, colam, args.mapping_loss_weight, args.alpha)      # TRAIN:363
    return total, out4, out


def train_step(model, guide, optimizer, batch, args: TrainArgs, ready=None, towers=None):
    """loss.backward(); optimizer.step(); scheduler.step(); zero_grad()  (TRAIN:364-374) — returns the device-side
    loss vector {total, txt, secla, colam} WITHOUT syncing (the reference's four .item() calls per step are gone)."""
    net = model.module if isinstance(model, DistributedDataParallel) else model
    if not net.training:
        net.train()
    ops.begin_step()                         # (a previous step that raised must not leak queued weight gradients into this one)
    total, out4, _ = forward_losses(model, guide, batch, args, ready, towers)
    with torch.autograd.set_multithreading_enabled(False):     # one device: the engine's worker-thread hop only costs host time
        # explicit unit gradient from a persistent constant: backward()'s default ones_like(total) is a fill kernel into a fresh
        # block per step — the one ATen kernel left inside the step (tools/aten_in_step.py), invisible to a launch plan
        total.backward(ops.const_one(total.device))
    ops.flush_wgrads()                       # normally empty (the engine's end-of-backward callback already ran)
    clip = None if args.no_clip_norm else args.clip_norm                         # TRAIN:365-366
    # native reducer: the bucket all-reduces sit ON the weight-gradient stream, and every bucket's AdamW range waits for that
    # bucket's own event — joining the whole stream here would hold AdamW back until the LAST collective has finished
    pipelined = isinstance(model, DistributedDataParallel) and model.native is not None and model.active and clip is None
    streams.join_all(skip_wgrad=pipelined)   # side streams -> compute stream
    if _PLAN is not None and _PLAN.towers is not None:
        _PLAN.mark("backward_done")          # (the host can start the next step's guide graph behind this point)
    if isinstance(model, DistributedDataParallel):
        model.reduce_and_step(optimizer, clip)       # all-reduce tail overlapped with the optimizer of the finished buckets
        if pipelined:
            streams.release_keep()           # the last bucket's event is behind everything the weight-gradient stream was given
    else:
        optimizer.step(clip_norm=clip)
    optimizer.zero_grad()
    return out4


def _model_inputs(net, batch, graphed=False):
    """masks + frozen-tower features for one batch, as every reference loop builds them (TRAIN:267-276,408-421,491-504).
    graphed: the image tower as a hipGraph per batch shape (caption generation at batch 1: ~250 launches of microseconds each)."""
    cfg = net.config
    src = batch["article_ids"]
    src_mask, _ = K.prep_ids(src, cfg.pad_token_id)
    if graphed:
        from .models.clip_vit import graphed_clip_img_feat
        feats = graphed_clip_img_feat(net.clip_model)(batch["img_tensor"])[image_feature_index(cfg)]
    else:
        feats = extract_clip_img_feat(net.clip_model, batch["img_tensor"])[image_feature_index(cfg)]
    kw = {}
    if not cfg.only_image:
        names_mask, _ = K.prep_ids(batch["names_art_ids"], cfg.pad_token_id)
        kw = dict(face_features=batch["face_emb"], face_mask=K.face_mask(batch["face_emb"]), name_ids=batch["names_art_ids"],
                  name_mask=names_mask)
    return src, src_mask, feats, kw


@torch.no_grad()
def eval_epoch(model, batches, device="cuda"):
    """TRAIN:391-447 / TRAINV:203-250: teacher-forced validation pass in eval mode.  Returns (mean of the per-batch text
    cross-entropy, out_dict) with out_dict[step] = {"logit_output": argmax token ids per sample, "gt_cap": target ids} — the
    reference stores the same two things as decoded strings (tokenizers are outside SURVEY §8).  One host sync per batch
    (the `.item()` of TRAIN:441), like the reference."""
    net = model.module if isinstance(model, DistributedDataParallel) else model
    cfg = net.config
    was_training = net.training
    net.eval()
    val_loss, n, out_dict = 0.0, 0, {}
    for step, batch in enumerate(batches):
        batch = to_device(batch, device)
        src, src_mask, feats, kw = _model_inputs(net, batch)
        tgt = batch["caption_ids"]
        _, tgt_in = K.prep_ids(tgt, cfg.pad_token_id, start_id=cfg.eos_token_id)
        out = net(input_ids=src, attention_mask=src_mask, decoder_input_ids=tgt_in, image_features=feats, labels=tgt,
                  output_logits=True, add_ner_ffn=True, **kw)
        lg = out["logits"]
        ids = K.argmax_rows(lg.view(-1, lg.shape[-1]), net.V).view(tgt.shape)
        out_dict[step] = {"logit_output": ids.tolist(), "gt_cap": tgt.tolist()}
        val_loss += float(out["loss"].item())
        n += 1
    net.train(was_training)
    return val_loss / max(n, 1), out_dict


@torch.no_grad()
def gen_caption_from_loader_bart(model, batches, beam_size, max_length, device="cuda", length_penalty=1.0, plm_type=None, pipeline=True,
                                 **gen_kw):
    """TRAIN:480-530 (the generation half; BLEU/ROUGE/CIDEr/METEOR scoring and detokenisation are outside SURVEY §8):
    out_dict[step] = {"gt": target ids, "gen": generated ids} with `model.generate(num_beams=beam_size, max_length=max_length)`;
    `length_penalty` is the extra knob of the stand-alone generator (DDPINF:38,867).  Arguments the reference leaves to the model's
    config (no_repeat_ngram_size, early_stopping, forced BOS/EOS) default to the hub checkpoint's (config.HUB_GENERATION_DEFAULTS,
    keyed by `plm_type`); `gen_kw` overrides them.  pipeline: overlap the next caption's encoder side with this caption's beam search."""
    net = model.module if isinstance(model, DistributedDataParallel) else model
    was_training = net.training
    net.eval()
    out_dict = {}
    from .config import generation_defaults
    gkw = dict(generation_defaults(plm_type), **gen_kw)        # what `model.generate(num_beams, max_length)` inherits from the hub config
    if pipeline and str(device).startswith("cuda"):
        # two-stage pipeline over the captions (generate.CaptionPipeline): image tower + encoder + cross K/V of caption i + 1 on a side
        # stream while caption i is decoded; same ids as the loop below
        from .generate import CaptionPipeline
        pipe = CaptionPipeline(net, lambda b: _model_inputs(net, b, graphed=True), beam_size, max_length=max_length,
                               length_penalty=length_penalty, add_ner_ffn=True, **gkw)
        with torch.no_grad():
            for step, (batch, gen) in enumerate(pipe(to_device(b, device) for b in batches)):
                out_dict[step] = {"gt": batch["caption_ids"].tolist(), "gen": gen.tolist()}
        net.train(was_training)
        return out_dict
    for step, batch in enumerate(batches):
        batch = to_device(batch, device)
        src, src_mask, feats, kw = _model_inputs(net, batch)
        gen = net.generate(input_ids=src, attention_mask=src_mask, num_beams=beam_size, max_length=max_length, image_features=feats,
                           length_penalty=length_penalty, add_ner_ffn=True, **kw, **gkw)
        out_dict[step] = {"gt": batch["caption_ids"].tolist(), "gen": gen.tolist()}
    net.train(was_training)
    return out_dict


class GraphedTrainStep:
    """The whole training step (all streams: compute, guide, weight gradients) captured once into a hipGraph and
    replayed per batch — "HIP graphs instead of a tracing compiler".  A step is ~2400 kernel launches that eager Python
    issues in ~31-37 ms; on this stack a replayed node costs about what an eager launch costs on the GPU side and the
    graph runs its parallel branches less concurrently, so replay is slower than eager multi-stream launches while the GPU
    needs ~74 ms per step: opt-in (`bench.py --graph`), kept working by tests/test_model_gpu.py.

    Capture-safety of the step: no host<->device sync inside it, LR / step counter / dropout counter live in device
    memory (lr_step kernel), every kernel is launched on torch's current stream (the capture stream or a side stream
    forked from it by an event), and all scratch comes from torch's graph-private allocator pool.
    Inputs are copied into static device buffers before each replay (a few hundred KiB of ids + the image batch).
    Not used with world_size > 1 (RCCL collectives stay outside a captured graph in round 1)."""

    def __init__(self, model, guide, optimizer, args: TrainArgs, example_batch, warmup=2):
        self.static = {k: v.clone() for k, v in example_batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # eager warm-up on a side stream (allocator + lazy inits)
            for _ in range(warmup):
                train_step(model, guide, optimizer, self.static, args)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out4 = train_step(model, guide, optimizer, self.static, args)

    def __call__(self, batch):
        for k, v in self.static.items():
            v.copy_(batch[k], non_blocking=True)
        self.graph.replay()
        return self.out4


class PlannedTrainStep:
    """The whole training step — frozen towers, forward, losses, backward, AdamW; every stream — as a LAUNCH PLAN: the step's
    C-ABI call sequence (kernel launches AND the fences between streams) is recorded once by the library while one step runs
    eagerly, and every later step is ONE call, vacnic_plan_replay, which re-issues the same launches on the same streams in the
    same order from C++ (include/vacnic_hip.h "launch plans").  The GPU-side schedule is the eager multi-stream one (a hipGraph
    of the step, GraphedTrainStep, runs its branches less concurrently and is slower); the ~1250 Python -> ctypes -> autograd
    round trips per step are gone from the launch path.

    What makes a step replayable: explicit scheduling (streams.explicit(): no stream synchronisation hidden inside torch), no
    host<->device sync and no ATen kernel inside the step, per-step scalars in device memory (LR, step and dropout counters:
    vacnic_lr_step), and every buffer at its recorded address — the recording runs inside a private allocator pool that is
    kept, inputs are copied into static tensors before each replay.  world_size > 1: on the reducer's native path the bucket
    all-reduces, their events and the waits before each bucket's AdamW range are C-ABI calls and hence plan commands; on the
    torch.distributed fallback they are host actions at marks of the plan (ddp.HOST_HOOK -> self.host), run unrecorded while the
    recording is paused and repeated by the host at the same points of every replay."""

    def __init__(self, model, guide, optimizer, args: TrainArgs, example_batch, warmup=2, towers=None):
        """towers: a FrozenTowerGraphs — the two frozen networks then stay hipGraph replays on their own streams, launched by the
        host before every plan replay (so they overlap the previous step's tail), and the plan is split at two marks where the
        host joins them (before the CoLaM loss) and releases their static outputs (after it)."""
        global _PLAN
        from . import _lib
        from . import ddp as _ddp
        if streams.enabled() and not streams.explicit():
            raise RuntimeError("PlannedTrainStep needs explicit scheduling (VACNIC_EXPLICIT_STREAMS=0 is set)")
        self.towers = towers
        self.ev_bwd = None
        self.static = {k: v.clone() for k, v in example_batch.items()}
        for _ in range(warmup):                              # eager: lazy buffers, kernel loading, allocator warm-up
            train_step(model, guide, optimizer, self.static, args, None, towers)
        torch.cuda.synchronize()
        self.stream = K._stream()
        self.pool = torch.cuda.MemPool()
        self.marks = []
        self._before()                                       # (host-side work of a step: not part of the plan)
        self.handle = int(_lib.lib.vacnic_plan_begin())
        if self.handle < 0:
            _lib.check(1)
        _PLAN = self
        # world > 1: the reducer's host-side work (bucket launches on the comm stream as backward completes them, the waits before
        # each bucket's AdamW range) becomes host actions at marks of the plan — N ranks then run the SAME launch path as one
        _ddp.HOST_HOOK = self.host
        try:
            with torch.cuda.use_mem_pool(self.pool):
                self.out4 = train_step(model, guide, optimizer, self.static, args, None, towers)
        except BaseException:
            # a recording that raised leaves neither a half-recorded plan nor its private pool behind
            _PLAN = None
            _ddp.HOST_HOOK = None
            _lib.lib.vacnic_plan_end(self.handle)
            _lib.lib.vacnic_plan_destroy(self.handle)
            self.handle = None
            self.pool = None
            raise
        _PLAN = None
        _ddp.HOST_HOOK = None
        _lib.check(_lib.lib.vacnic_plan_end(self.handle))
        torch.cuda.synchronize()
        self.commands = int(_lib.lib.vacnic_plan_size(self.handle))

    def mark(self, what):
        """called by forward_losses while recording: the host acts at this point of every replay."""
        from . import _lib
        self.marks.append((int(_lib.lib.vacnic_plan_mark()), what))
        self._at(what)

    def host(self, fn):
        """recording: register `fn` as a host action at this point of the plan and run it now, unrecorded (whatever it launches
        through the C-ABI is the host's to repeat at every replay, not the plan's)."""
        from . import _lib
        self.marks.append((int(_lib.lib.vacnic_plan_mark()), fn))
        _lib.check(_lib.lib.vacnic_plan_pause(1))
        try:
            fn()
        finally:
            _lib.check(_lib.lib.vacnic_plan_pause(0))

    def _before(self, batch=None, ready=None):
        if self.towers is not None:
            late = self.ev_bwd if __import__("os").environ.get("VACNIC_GUIDE_LATE", "0") == "1" else None
            self.towers.launch_graphs(batch if batch is not None else self.static, ready, guide_after=late)
            torch.cuda.current_stream().wait_event(self.towers.ev_vit)         # the student's encoder needs the image feature

    def _at(self, what):
        if callable(what):
            what()
        elif what == "join_guide":
            torch.cuda.current_stream().wait_stream(streams.aux_stream())
        elif what == "towers_consumed":
            self.towers.mark_consumed()
        elif what == "backward_done":
            self.ev_bwd = torch.cuda.Event()
            self.ev_bwd.record()

    def __call__(self, batch, ready=None):
        """ready: optional event after which `batch` is resident in HBM (the tower graphs then start on it instead of behind the
        previous step on the compute stream, like train_step's `ready`)."""
        from . import _lib
        if K._stream() != self.stream:
            raise RuntimeError("PlannedTrainStep: replay on the stream the plan was recorded on")
        self._before(batch, ready)
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)
        for k, v in self.static.items():
            if batch[k] is not v:
                v.copy_(batch[k], non_blocking=True)
        pos = 0
        for idx, what in self.marks:
            _lib.call("vacnic_plan_replay", self.handle, pos, idx)
            self._at(what)
            pos = idx
        _lib.call("vacnic_plan_replay", self.handle, pos, self.commands)
        return self.out4

    def close(self):
        from . import _lib
        if self.handle is not None:
            _lib.check(_lib.lib.vacnic_plan_destroy(self.handle))
            self.handle = None


class FrozenTowerGraphs:
    """hipGraph replay for the two frozen, autograd-free, static-shape networks of the step (guide BART forward and
    CLIP ViT forward): ~430 of the step's launches become two graph launches on their side streams, which takes them off
    the Python launch path while the trainable network keeps eager multi-stream launches
    (a single whole-step graph measured slower: hipGraph runs its parallel branches less concurrently than streams do).

    Static buffers: inputs are copied in on the tower's stream right before the replay; the outputs (`gh`, `img_cls`) are
    read by the main chain only during the forward pass, so the next replay waits for the `consumed` event recorded
    after the CoLaM forward (write-after-read), and the id masks the backward needs are fresh tensors every step."""

    def __init__(self, net, guide, example_batch):
        cfg = net.config
        self.net, self.guide = net, guide
        aux, vis = streams.aux_stream(), streams.vit_stream()
        if aux is None:
            raise RuntimeError("FrozenTowerGraphs needs streams.enable(True)")
        torch.cuda.synchronize()
        self.pad, self.start = cfg.pad_token_id, cfg.eos_token_id
        self.src_s = example_batch["article_ids"].clone()
        self.img_s = example_batch["img_tensor"].clone()
        self.mask_s, _ = K.prep_ids(self.src_s, self.pad)
        _, self.tgtin_s = K.prep_ids(example_batch["caption_ids"].clone(), self.pad, start_id=self.start)
        self.consumed = None
        self.ev_vit = None
        self.gh = self.g_guide = None

        def guide_body():
            return guide(input_ids=self.src_s, attention_mask=self.mask_s, decoder_input_ids=self.tgtin_s)["decoder_hidden_states"][-1]

        def vit_body():
            return extract_clip_img_feat(net.clip_model, self.img_s)[image_feature_index(cfg)]

        torch.cuda.synchronize()
        with torch.no_grad():
            with torch.cuda.stream(vis):                     # eager warm-up on the capture streams
                vit_body()
            if guide is not None:
                with torch.cuda.stream(aux):
                    guide_body()
            torch.cuda.synchronize()
            if guide is not None:
                self.g_guide = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_guide, stream=aux):
                    self.gh = guide_body()
            self.g_vit = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_vit, stream=vis):
                self.img_cls = vit_body()
        torch.cuda.synchronize()

    def launch(self, batch, ready):
        """enqueue id preprocessing (eager, fresh tensors) and both tower replays on their streams."""
        aux, vis = streams.aux_stream(), streams.vit_stream()
        main = torch.cuda.current_stream()
        for s_ in (aux, vis):
            if ready is None:
                s_.wait_stream(main)
            else:
                s_.wait_event(ready)
            if self.consumed is not None:
                s_.wait_event(self.consumed)
        src, tgt = batch["article_ids"], batch["caption_ids"]
        with torch.cuda.stream(aux):
            src_mask, _ = K.prep_ids(src, self.pad)
            tgt_mask, tgt_in = K.prep_ids(tgt, self.pad, start_id=self.start)
            ev_prep = torch.cuda.Event()
            ev_prep.record(aux)
        with torch.cuda.stream(vis):                 # the student's encoder input needs the image feature: ViT goes first
            self.img_s.copy_(batch["img_tensor"], non_blocking=True)
            self.g_vit.replay()
            self.ev_vit = torch.cuda.Event()
            self.ev_vit.record(vis)                  # (the towers may share one stream: wait for THIS, not for the stream's tail)
        if self.g_guide is not None:
            with torch.cuda.stream(aux):             # the guide's output is needed only by the CoLaM loss
                self.src_s.copy_(src, non_blocking=True)
                self.mask_s.copy_(src_mask, non_blocking=True)
                self.tgtin_s.copy_(tgt_in, non_blocking=True)
                self.g_guide.replay()
        for tns in (src, tgt):
            tns.record_stream(aux)
        batch["img_tensor"].record_stream(vis)
        return src_mask, tgt_mask, tgt_in, ev_prep

    def launch_graphs(self, batch, ready=None, guide_after=None):
        """both tower replays only (a launch plan computes the id masks itself): the guide's id inputs are derived here on its own
        stream from the batch."""
        aux, vis = streams.aux_stream(), streams.vit_stream()
        main = torch.cuda.current_stream()
        for s_ in (aux, vis):
            # the towers read the CALLER's batch (resident: `ready`, or ordered on the compute stream), not the plan's static copy,
            # so they need not queue behind the previous step's tail on the compute stream
            if ready is None:
                s_.wait_stream(main)
            else:
                s_.wait_event(ready)
            if self.consumed is not None:
                s_.wait_event(self.consumed)
        with torch.cuda.stream(vis):
            self.img_s.copy_(batch["img_tensor"], non_blocking=True)
            self.g_vit.replay()
            self.ev_vit = torch.cuda.Event()
            self.ev_vit.record(vis)
        if self.g_guide is not None:
            with torch.cuda.stream(aux):
                if guide_after is not None:
                    aux.wait_event(guide_after)      # the guide (needed only at the CoLaM loss) runs beside the previous AdamW, not its backward
                self.src_s.copy_(batch["article_ids"], non_blocking=True)
                K.prep_ids_into(self.src_s, self.mask_s, None, self.pad)
                K.prep_ids_into(batch["caption_ids"], None, self.tgtin_s, self.pad, self.start)
                self.g_guide.replay()

    def mark_consumed(self):
        self.consumed = torch.cuda.Event()
        self.consumed.record(torch.cuda.current_stream())
