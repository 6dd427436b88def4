"""Entry point with the file name and flags of the reference's image-only trainer (TRAINV:1-120, run_onlyvis_train.sh) for the
MI355X-native step: BART + CLIP ViT with the visual prompt only (`--only_image True`), text cross-entropy as the whole loss
(TRAINV:171-181 — no guide network, no face/name branch, no CoLaM / SECLA terms).

    torchrun --nproc_per_node=N run_train_mmbart_enc_self_onlyvis_retrieve_crossattn.py --plm_type facebook/bart-base \
        --clip_type ViT-B/16 --enc_fusion_layer 0 1 ... 11 --only_image True --no_mapping True --use_secla False ...

Same step, same scope notes and same `run()` as the full trainer's entry point next to this file; this one only differs in the
flag defaults the reference's image-only script has (`--only_image`, `--do_retrieval`).
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
_spec = importlib.util.spec_from_file_location(
    "vacnic_full_trainer", os.path.join(_HERE, "train_mmbart_enc_self_face_name_ids_retrieve_crossattn_bart_guide_match.py"))
_full = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_full)

parser = _full.parser
parser.add_argument("--do_retrieval", action="store_true")        # TRAINV: retrieved-sentence articles; a data-side switch
parser.set_defaults(only_image=True, no_mapping=True, use_secla=False, plm_type="facebook/bart-base", clip_type="ViT-B/16")
run = _full.run


if __name__ == "__main__":
    args = parser.parse_args()
    if not args.only_image:
        raise ValueError("the image-only trainer builds the image-only model (TRAINV:538): pass --only_image True, or use "
                         "train_mmbart_enc_self_face_name_ids_retrieve_crossattn_bart_guide_match.py for the full model")
    run(args)
