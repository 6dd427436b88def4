"""CLIP visual tower (frozen) on gfx950 kernels, with openai-CLIP parameter names
(`visual.conv1.weight`, `visual.transformer.resblocks.N.attn.in_proj_weight`, ...; clip==1.0 model.py),
driven exactly like the reference does at TRAIN:220-240 (`extract_clip_img_feat`): conv1 patch-embed
(kernel = stride = patch, no bias) -> [cls; patches] + pos -> ln_pre -> L x {x += MHA(ln_1 x);
x += c_proj(QuickGELU(c_fc(ln_2 x)))} -> ln_post on CLS and on patches, no `proj`, no grad, fp32 out.

The patch embedding is an im2col (strided patch gather, bf16) followed by the MFMA GEMM; attention is the
fused kernel without mask; residual adds are fused into the out_proj / c_proj GEMM epilogues.
"""
import torch
from torch import nn

from .. import kernels as K
from .. import ops
from ..arena import ParamArena
from ..config import ClipVisionConfig
from ..ops import LinearSpec


class _MHA(nn.Module):
    """parameter container named like nn.MultiheadAttention (in_proj_weight [3w,w] in q,k,v order)."""

    def __init__(self, w):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * w, w))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * w))
        self.out_proj = nn.Linear(w, w)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.attn = _MHA(w)
        self.ln_1 = nn.LayerNorm(w)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(w, 4 * w))
        self.mlp.add_module("gelu", nn.Identity())           # QuickGELU lives in the GEMM epilogue
        self.mlp.add_module("c_proj", nn.Linear(4 * w, w))
        self.ln_2 = nn.LayerNorm(w)


class _Transformer(nn.Module):
    def __init__(self, w, layers):
        super().__init__()
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(w) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, vcfg: ClipVisionConfig):
        super().__init__()
        self.vcfg = vcfg
        w = vcfg.width
        self.conv1 = nn.Conv2d(3, w, vcfg.patch_size, vcfg.patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.empty(w))
        self.positional_embedding = nn.Parameter(torch.empty(vcfg.tokens, w))
        self.ln_pre = nn.LayerNorm(w)
        self.transformer = _Transformer(w, vcfg.layers)
        self.ln_post = nn.LayerNorm(w)
        self.proj = nn.Parameter(torch.empty(w, vcfg.output_dim))     # unused on this path (TRAIN:236 skips it)
        for p in self.parameters():
            p.requires_grad = False
        self.arena = None

    def finalize(self, device="cuda"):
        v = self.vcfg
        self.arena = ParamArena(self, device, trainable=False)
        K_real = 3 * v.patch_size * v.patch_size
        self.Kp = (K_real + 31) // 32 * 32
        # conv1 weight [w,3,p,p] -> GEMM weight [w, Kp] bf16, zero padded (frozen: built once)
        wmat = torch.zeros((v.width, self.Kp), device=device, dtype=torch.bfloat16)
        self.s_patch = LinearSpec(wmat)

        def repack():           # in place: tower hipGraphs captured earlier keep pointing at the same buffer
            wmat[:, :K_real] = self.conv1.weight.w16.reshape(v.width, K_real)
        repack()
        # the repack is a derived copy of conv1.weight: redo it whenever the arena's bf16 shadow is refreshed (weights drawn or
        # loaded after finalize) — without this the tower kept the constructor's random patch embedding
        self.arena.refresh_hooks.append(repack)
        self.blocks = []
        for blk in self.transformer.resblocks:
            self.blocks.append(dict(
                qkv=LinearSpec(blk.attn.in_proj_weight.w16, blk.attn.in_proj_bias.data),
                out=LinearSpec(blk.attn.out_proj.weight.w16, blk.attn.out_proj.bias.data),
                fc=LinearSpec(blk.mlp.c_fc.weight.w16, blk.mlp.c_fc.bias.data),
                proj=LinearSpec(blk.mlp.c_proj.weight.w16, blk.mlp.c_proj.bias.data),
                ln1=(blk.ln_1.weight.data, blk.ln_1.bias.data), ln2=(blk.ln_2.weight.data, blk.ln_2.bias.data)))
        return self

    @torch.no_grad()
    def features(self, img):
        """img fp32 [B,3,HW,HW] -> (patches [B,g*g,w], cls [B,w]) bf16, both after ln_post."""
        v = self.vcfg
        B = img.shape[0]
        w, T, H = v.width, v.tokens, v.heads
        patches = K.im2col_patches(img.contiguous().float(), v.patch_size, self.Kp)
        pe = K.gemm(patches, self.s_patch.w16, B * (T - 1), w, self.Kp)
        x = K.vit_assemble(pe, self.class_embedding.w16, self.positional_embedding.w16, B, T - 1, w)
        x, _, _ = K.add_ln_fwd(x, None, self.ln_pre.weight.data, self.ln_pre.bias.data, need_stats=False)
        M = B * T
        for b in self.blocks:
            y, _, _ = K.add_ln_fwd(x, None, b["ln1"][0], b["ln1"][1], need_stats=False)
            qkv = K.gemm(y.view(M, w), b["qkv"].w16, M, 3 * w, w, bias=b["qkv"].bias).view(B, T, 3 * w)
            ctx, _ = K.attn_fwd(qkv[..., :w], qkv[..., w:2 * w], qkv[..., 2 * w:], B, H, T, T, scale=0.125, need_lse=False)
            x = K.gemm(ctx.view(M, w), b["out"].w16, M, w, w, bias=b["out"].bias, residual=x.view(M, w)).view(B, T, w)
            y, _, _ = K.add_ln_fwd(x, None, b["ln2"][0], b["ln2"][1], need_stats=False)
            hmid = K.gemm(y.view(M, w), b["fc"].w16, M, 4 * w, w, bias=b["fc"].bias, act="quick_gelu")
            x = K.gemm(hmid, b["proj"].w16, M, w, 4 * w, bias=b["proj"].bias, residual=x.view(M, w)).view(B, T, w)
        out, _, _ = K.add_ln_fwd(x, None, self.ln_post.weight.data, self.ln_post.bias.data, need_stats=False)
        return out[:, 1:, :], out[:, 0, :]


class CLIPVisualOnly(nn.Module):
    """Stand-in for the `clip_model` object the reference passes around (TRAIN:737-743): only `.visual` is used
    on this path (`--no_clip_loss True --freeze_clip True`, run_full_train.sh:21,24)."""

    def __init__(self, vcfg: ClipVisionConfig):
        super().__init__()
        self.visual = VisionTransformer(vcfg)

    def finalize(self, device="cuda"):
        self.visual.finalize(device)
        return self


def extract_clip_img_feat(clip_model, x):
    """Drop-in for TRAIN:220-240: returns (ln_post(patch tokens), ln_post(cls)) as fp32 tensors, no grad."""
    with torch.no_grad():
        if clip_model.training:
            clip_model.eval()
        vis = clip_model.visual
        if vis.arena is None:
            raise RuntimeError("call clip_model.finalize(device) first")
        patches, cls = vis.features(x)                    # strided views of one [B, 1 + g*g, w] tensor
        B, G2, w = patches.shape
        pc = torch.empty((B, G2, w), device=patches.device, dtype=patches.dtype)
        cc = torch.empty((B, 1, w), device=patches.device, dtype=patches.dtype)
        K.copy3d(patches, pc, B, G2, w)                    # (our strided-copy kernel, not Tensor.contiguous(): no ATen kernel on the path)
        K.copy3d(cls.unsqueeze(1), cc, B, 1, w)
        return K.cast_bf16_f32(pc), K.cast_bf16_f32(cc.view(B, w))


def graphed_clip_img_feat(clip_model):
    """extract_clip_img_feat as a hipGraph per image-batch shape (caption generation at batch 1: the ViT is ~250 launches of a few
    microseconds each); returns a callable x -> (patch tokens, cls).  The outputs are overwritten by the next call."""
    from ..generate import GraphedCall
    g = clip_model.__dict__.get("_graphed_feat")
    if g is None:
        g = clip_model.__dict__["_graphed_feat"] = GraphedCall(lambda x: extract_clip_img_feat(clip_model, x))
    return g
