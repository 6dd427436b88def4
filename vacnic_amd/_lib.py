"""ctypes binding of libvacnic_hip.so (the C-ABI declared in include/vacnic_hip.h).

This is the stub a maintainer of the reference would add: the reference has no
FFI layer of its own, its ops are torch.nn calls inside `src/models` (MFULL) and the trainer
(TRAIN).  There is NO CPU fallback: if the shared object is missing or a symbol is absent the
import fails loudly, and every call checks the returned status.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvacnic_hip.so")


class VacnicError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vacnic_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    return C.CDLL(LIB_PATH)


lib = _load()

i32, i64, f32, u64, vp = C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_void_p


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": fields})


GemmArgs = _struct("vacnic_gemm_args", [
    ("x", vp), ("w", vp), ("bias", vp), ("out", vp), ("preact", vp), ("dact_src", vp), ("residual", vp), ("xsum", vp),
    ("M", i64), ("N", i64), ("K", i64), ("ldx", i64), ("ldw", i64), ("ldo", i64),
    ("x_kstrided", i32), ("w_kstrided", i32), ("act", i32), ("out_mode", i32), ("split_k", i32), ("alpha", f32), ("tile_hint", i32),
    ("workspace", vp), ("workspace_bytes", i64), ("counters", vp), ("counters_len", i64),
    ("drop_p", f32), ("drop_seed", u64), ("drop_seed_dev", vp)])

GemvLnArgs = _struct("vacnic_gemv_ln_args", [
    ("x", vp), ("residual", vp), ("gamma", vp), ("beta", vp), ("ln_out", vp), ("w", vp), ("bias", vp), ("out", vp),
    ("M", i64), ("N", i64), ("K", i64), ("ldw", i64), ("ldo", i64), ("act", i32), ("out_mode", i32), ("eps", f32)])

WgradJob = _struct("vacnic_wgrad_job", [
    ("dy", vp), ("x", vp), ("dw", vp), ("dbias", vp), ("M", i64), ("N", i64), ("K", i64), ("lddy", i64), ("ldx", i64), ("lddw", i64)])

AttnFwdArgs = _struct("vacnic_attn_fwd_args", [
    ("q", vp), ("k", vp), ("v", vp), ("out", vp), ("lse", vp), ("key_mask", vp),
    ("B", i64), ("H", i64), ("Tq", i64), ("Tk", i64),
    ("ldq", i64), ("ldk", i64), ("ldv", i64), ("ldo", i64),
    ("bsq", i64), ("bsk", i64), ("bsv", i64), ("bso", i64),
    ("causal", i32), ("scale", f32), ("p_drop", f32), ("seed", u64), ("seed_dev", vp)])

AttnBwdArgs = _struct("vacnic_attn_bwd_args", [
    ("q", vp), ("k", vp), ("v", vp), ("out", vp), ("dout", vp), ("lse", vp), ("delta", vp),
    ("dq", vp), ("dk", vp), ("dv", vp), ("key_mask", vp),
    ("B", i64), ("H", i64), ("Tq", i64), ("Tk", i64),
    ("ldq", i64), ("ldk", i64), ("ldv", i64), ("ldo", i64),
    ("bsq", i64), ("bsk", i64), ("bsv", i64), ("bso", i64),
    ("lddq", i64), ("lddk", i64), ("lddv", i64), ("bsdq", i64), ("bsdk", i64), ("bsdv", i64),
    ("causal", i32), ("scale", f32), ("p_drop", f32), ("seed", u64), ("seed_dev", vp)])

AddLnFwdArgs = _struct("vacnic_add_ln_fwd_args", [
    ("x", vp), ("residual", vp), ("gamma", vp), ("beta", vp), ("out", vp), ("mean", vp), ("rstd", vp),
    ("R", i64), ("D", i64), ("eps", f32), ("p_drop", f32), ("seed", u64), ("seed_dev", vp)])

AddLnBwdArgs = _struct("vacnic_add_ln_bwd_args", [
    ("dout", vp), ("x", vp), ("residual", vp), ("gamma", vp), ("mean", vp), ("rstd", vp),
    ("dresidual", vp), ("dx", vp), ("dgamma", vp), ("dbeta", vp),
    ("R", i64), ("D", i64), ("p_drop", f32), ("seed", u64), ("seed_dev", vp), ("partials", vp), ("partial_rows", i64),
    ("defer_fold", i32)])

EmbedLnFwdArgs = _struct("vacnic_embed_ln_fwd_args", [
    ("ids", vp), ("embed", vp), ("pos", vp), ("gamma", vp), ("beta", vp), ("out", vp), ("mean", vp), ("rstd", vp),
    ("B", i64), ("T", i64), ("D", i64), ("V", i64), ("pos_offset", i64),
    ("embed_scale", f32), ("eps", f32), ("p_drop", f32), ("seed", u64), ("seed_dev", vp)])

EmbedLnBwdArgs = _struct("vacnic_embed_ln_bwd_args", [
    ("ids", vp), ("embed", vp), ("pos", vp), ("dout", vp), ("gamma", vp), ("mean", vp), ("rstd", vp),
    ("dembed", vp), ("dpos", vp), ("dgamma", vp), ("dbeta", vp),
    ("B", i64), ("T", i64), ("D", i64), ("V", i64), ("pos_offset", i64), ("embed_scale", f32),
    ("padding_idx", i64), ("p_drop", f32), ("seed", u64), ("seed_dev", vp)])

CeArgs = _struct("vacnic_ce_args", [
    ("logits", vp), ("targets", vp), ("row_lse", vp), ("row_loss", vp), ("loss_sum", vp), ("count", vp),
    ("dlogits", vp), ("grad_out", vp), ("grad_scale", f32),
    ("R", i64), ("V", i64), ("ldl", i64), ("ldd", i64), ("ignore_index", i64), ("logits_f32", i32)])

ColamFwdArgs = _struct("vacnic_colam_fwd_args", [
    ("hs", vp), ("hg", vp), ("mask", vp), ("loss", vp), ("cos", vp), ("pooled_s", vp), ("pooled_g", vp),
    ("B", i64), ("T", i64), ("D", i64), ("margin", f32)])

ColamBwdArgs = _struct("vacnic_colam_bwd_args", [
    ("cos", vp), ("pooled_s", vp), ("pooled_g", vp), ("mask", vp), ("dhs", vp),
    ("B", i64), ("T", i64), ("D", i64), ("margin", f32), ("grad_out", vp), ("grad_scale", f32)])

SeclaFwdArgs = _struct("vacnic_secla_fwd_args", [
    ("faces", vp), ("names", vp), ("sim", vp), ("logits1", vp), ("logits2", vp), ("loss", vp),
    ("B", i64), ("F", i64), ("N", i64), ("D", i64)])

SeclaBwdArgs = _struct("vacnic_secla_bwd_args", [
    ("faces", vp), ("names", vp), ("sim", vp), ("logits1", vp), ("logits2", vp), ("dfaces", vp), ("wsim", vp),
    ("B", i64), ("F", i64), ("N", i64), ("D", i64), ("grad_out", vp), ("grad_scale", f32)])

NameEmbedArgs = _struct("vacnic_name_embed_args", [
    ("ids", vp), ("embed", vp), ("pos", vp), ("gamma", vp), ("beta", vp), ("out", vp),
    ("B", i64), ("Nn", i64), ("Ln", i64), ("D", i64), ("V", i64), ("pos_offset", i64),
    ("embed_scale", f32), ("eps", f32)])

LmheadCeArgs = _struct("vacnic_lmhead_ce_args", [
    ("h", vp), ("emb", vp), ("bias", vp), ("targets", vp), ("part", vp), ("tl", vp), ("row_lse", vp), ("loss_sum", vp), ("count", vp),
    ("R", i64), ("V", i64), ("D", i64), ("ldh", i64), ("lde", i64), ("part_tiles", i64), ("ignore_index", i64)])

BeamState = _struct("vacnic_beam_state", [
    ("seq0", vp), ("seq1", vp), ("beam_scores", vp), ("done", vp), ("hyp_cnt", vp), ("hyp_worst", vp), ("hyp_score", vp), ("hyp_len", vp),
    ("hyp_seq", vp), ("next_ids", vp), ("src_idx", vp), ("bans", vp),
    ("B", i64), ("nb", i64), ("Lmax", i64), ("V", i64), ("eos", i64), ("pad", i64), ("no_repeat_ngram_size", i64), ("early_stopping", i64),
    ("length_penalty", f32)])

LmheadTopkArgs = _struct("vacnic_lmhead_topk_args", [
    ("h", vp), ("emb", vp), ("bias", vp), ("logits", vp), ("workspace", vp), ("beam_scores", vp), ("bans", vp), ("top_val", vp), ("top_idx", vp),
    ("R", i64), ("V", i64), ("d", i64), ("ldw", i64), ("ldl", i64), ("workspace_floats", i64),
    ("n_ban", i32), ("eos", i32), ("suppress_eos", i32), ("forced_token", i32), ("K2", i32)])

DecoderLayer = _struct("vacnic_decoder_layer", [
    ("w_kvq", vp), ("w_so", vp), ("w_cq", vp), ("w_co", vp), ("w_fc1", vp), ("w_fc2", vp),
    ("b_kvq", vp), ("b_so", vp), ("b_cq", vp), ("b_co", vp), ("b_fc1", vp), ("b_fc2", vp),
    ("ln_self_g", vp), ("ln_self_b", vp), ("ln_cross_g", vp), ("ln_cross_b", vp), ("ln_final_g", vp), ("ln_final_b", vp),
    ("cross_kv", vp), ("cross_bs", i64)])

DecoderStepArgs = _struct("vacnic_decoder_step_args", [
    ("layers", vp), ("cache", vp), ("h0", vp), ("hbuf0", vp), ("hbuf1", vp), ("obuf", vp), ("ctx", vp), ("qbuf", vp), ("fbuf", vp),
    ("enc_mask", vp), ("sync", vp), ("slots", vp),
    ("L", i64), ("R", i64), ("d", i64), ("H", i64), ("F", i64), ("S", i64), ("t", i64), ("Tmax", i64), ("eps", f32), ("scale", f32),
    ("trace", vp), ("trace_wg", i64)])

AdamwArgs = _struct("vacnic_adamw_args", [
    ("p", vp), ("g", vp), ("m", vp), ("v", vp), ("p_bf16", vp), ("hyper", vp),
    ("n", i64), ("beta1", f32), ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("grad_scale", f32),
    ("zero_grad", i32), ("clip_coef", vp)])

# symbol -> argtypes.  EVERY function include/vacnic_hip.h declares must appear here
# (tests/test_abi.py parses the header and checks both directions).
_STRUCT_FNS = {
    "vacnic_gemm_bf16": GemmArgs, "vacnic_gemv_ln_bf16": GemvLnArgs, "vacnic_attn_fwd": AttnFwdArgs, "vacnic_attn_bwd": AttnBwdArgs,
    "vacnic_add_ln_fwd": AddLnFwdArgs, "vacnic_add_ln_bwd": AddLnBwdArgs,
    "vacnic_embed_ln_fwd": EmbedLnFwdArgs, "vacnic_embed_ln_bwd": EmbedLnBwdArgs,
    "vacnic_ce_fwd": CeArgs, "vacnic_ce_bwd": CeArgs,
    "vacnic_colam_fwd": ColamFwdArgs, "vacnic_colam_bwd": ColamBwdArgs,
    "vacnic_secla_fwd": SeclaFwdArgs, "vacnic_secla_bwd": SeclaBwdArgs,
    "vacnic_name_embed_mean": NameEmbedArgs, "vacnic_adamw": AdamwArgs, "vacnic_lmhead_ce_fwd": LmheadCeArgs,
    "vacnic_decoder_step": DecoderStepArgs, "vacnic_lmhead_topk": LmheadTopkArgs,
}
_PLAIN_FNS = {
    "vacnic_combine_losses": [vp, vp, vp, vp, f32, f32, vp, vp],
    "vacnic_lr_step": [vp, f32, f32, f32, vp, vp],
    "vacnic_grad_clip_coef": [vp, i64, f32, f32, vp, vp, vp],
    "vacnic_cast_f32_bf16": [vp, vp, i64, vp],
    "vacnic_cast_bf16_f32": [vp, vp, i64, vp],
    "vacnic_copy2d_bf16": [vp, vp, i64, i64, i64, i64, i32, vp],
    "vacnic_copy3d_bf16": [vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, vp],
    "vacnic_pad_cols_bf16": [vp, vp, i64, i64, i64, i64, vp],
    "vacnic_im2col_patches": [vp, vp, i64, i64, i64, i64, vp],
    "vacnic_vit_assemble": [vp, vp, vp, vp, i64, i64, i64, vp],
    "vacnic_prep_ids": [vp, vp, vp, i64, i64, i64, i64, vp],
    "vacnic_face_mask": [vp, vp, i64, i64, vp],
    "vacnic_cat2_u8": [vp, vp, vp, i64, i64, i64, vp],
    "vacnic_dropout_bf16": [vp, vp, i64, f32, u64, vp, vp],
    "vacnic_argmax_rows": [vp, vp, i64, i64, i64, i32, vp],
    "vacnic_bias_grad": [vp, vp, i64, i64, i64, vp],
    "vacnic_add_bf16": [vp, vp, vp, i64, vp],
    "vacnic_probe_layouts": [vp, vp, i64, vp],
    "vacnic_beam_topk": [vp, vp, vp, i32, i32, i32, i32, vp, vp, i64, i64, i64, i32, i32, vp],
    "vacnic_gather_rows": [vp, vp, vp, i64, i64, i64, i64, vp],
    "vacnic_image_u8_normalize": [vp, vp, vp, i64, i64, i64, f32, f32, f32, f32, f32, f32, vp],
    "vacnic_lmhead_ce_rowp": [vp, vp, vp, vp, f32, vp, i64, i64, vp],
    "vacnic_lmhead_ce_dlogits": [C.POINTER(LmheadCeArgs), i64, i64, vp, i64, vp, vp],
    "vacnic_zero_bytes": [vp, i64, vp],
    "vacnic_beam_init": [C.POINTER(BeamState), i32, vp],
    "vacnic_wgrad_group": [C.POINTER(WgradJob), i64, vp],
    "vacnic_plan_end": [i64], "vacnic_plan_replay": [i64, i64, i64], "vacnic_plan_destroy": [i64], "vacnic_stream_fence": [vp, vp],
    "vacnic_plan_pause": [i32],
    "vacnic_ln_partial_fold": [vp, vp, vp, i64, i64, vp],
    "vacnic_comm_load": [C.c_char_p], "vacnic_comm_unique_id": [vp], "vacnic_allreduce_bucket": [i64, vp, i64, i32, vp],
    "vacnic_comm_broadcast": [i64, vp, i64, i32, i32, vp], "vacnic_comm_destroy": [i64],
    "vacnic_event_record": [i32, vp], "vacnic_event_wait": [i32, vp],
    "vacnic_beam_step": [C.POINTER(BeamState), vp, vp, i32, i32, vp],
}
EXPORTED = sorted(list(_STRUCT_FNS) + list(_PLAIN_FNS) + ["vacnic_last_error_string", "vacnic_version", "vacnic_decoder_step_sync_bytes", "vacnic_decoder_step_slots_bytes",
                                                            "vacnic_plan_begin", "vacnic_plan_size", "vacnic_plan_mark",
                                                            "vacnic_gemm_workspace_bytes", "vacnic_gemm_counters", "vacnic_comm_init",
                                                            "vacnic_lmhead_topk_workspace"])

for _name, _st in _STRUCT_FNS.items():
    _fn = getattr(lib, _name)          # AttributeError here = stale .so: fail loudly
    _fn.argtypes = [C.POINTER(_st), vp]
    _fn.restype = C.c_int
for _name, _at in _PLAIN_FNS.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _at
    _fn.restype = C.c_int
lib.vacnic_last_error_string.restype = C.c_char_p
lib.vacnic_last_error_string.argtypes = []
lib.vacnic_version.restype = C.c_int
lib.vacnic_version.argtypes = []
lib.vacnic_decoder_step_sync_bytes.restype = C.c_int64
lib.vacnic_decoder_step_sync_bytes.argtypes = []
lib.vacnic_decoder_step_slots_bytes.restype = C.c_int64
lib.vacnic_decoder_step_slots_bytes.argtypes = [C.c_int64]
lib.vacnic_plan_begin.restype = C.c_int64
lib.vacnic_plan_begin.argtypes = []
lib.vacnic_plan_size.restype = C.c_int64
lib.vacnic_plan_size.argtypes = [C.c_int64]
lib.vacnic_plan_mark.restype = C.c_int64
lib.vacnic_plan_mark.argtypes = []
lib.vacnic_lmhead_topk_workspace.restype = C.c_int64
lib.vacnic_lmhead_topk_workspace.argtypes = [C.c_int64, C.c_int64, C.c_int32]
lib.vacnic_gemm_workspace_bytes.restype = C.c_int64
lib.vacnic_gemm_workspace_bytes.argtypes = [C.c_int64, C.c_int64, C.c_int64]
lib.vacnic_gemm_counters.restype = C.c_int64
lib.vacnic_gemm_counters.argtypes = [C.c_int64, C.c_int64]
lib.vacnic_comm_init.restype = C.c_int64
lib.vacnic_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32]

_VALUE_ERRORS = (1, 2, 3)   # bad shape / dtype / alignment -> ValueError like the reference's shape checks


def check(status):
    if status != 0:
        msg = lib.vacnic_last_error_string().decode()
        if status in _VALUE_ERRORS:
            raise ValueError(f"vacnic_hip: {msg}")
        raise VacnicError(f"vacnic_hip status {status}: {msg}")


CALLS = 0          # number of C-ABI calls issued so far (bench.py reports calls per step)


def call_struct(name, **kw):
    """Call a struct-taking entry point: call_struct('vacnic_gemm_bf16', stream=s, x=ptr, ...)."""
    global CALLS
    CALLS += 1
    stream = kw.pop("stream")
    st = _STRUCT_FNS[name](**kw)
    check(getattr(lib, name)(C.byref(st), stream))


def call(name, *args):
    global CALLS
    CALLS += 1
    check(getattr(lib, name)(*args))
