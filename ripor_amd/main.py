"""``python -m t5_pretrainer.main`` for the one training task on this repository's path: the prefix-oriented ranking
fine-tune (``--loss_type=t5seq_aq_encoder_lng_knp_margin_mse``; reference main.py:68-74, 93-94, 127-186 and
full_scripts/full_lng_knp_train_pipline.sh:80-99). Same flags as the reference's ``Arguments`` as far as that task reads
them; every other loss type is out of scope and refused.

One process per GPU (``torchrun --nproc-per-node N -m t5_pretrainer.main ...``): ``torch.distributed`` backend "nccl" = RCCL;
the gradient exchange is bucketed and overlapped with the backward inside ``training_step``."""
from __future__ import annotations

import argparse
import json
import os

import torch


def get_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--loss_type", default="t5seq_aq_encoder_lng_knp_margin_mse")
    ap.add_argument("--model_type", default="t5_docid_gen_encoder")
    ap.add_argument("--model_name_or_path", default="t5-base", help="tokenizer source (a directory works offline)")
    ap.add_argument("--pretrained_path", required=True)
    ap.add_argument("--teacher_score_path", required=True)
    ap.add_argument("--collection_path", default=None)
    ap.add_argument("--queries_path", required=True)
    ap.add_argument("--docid_to_smtid_path", default=None)
    ap.add_argument("--smtid_as_docid", action="store_true")
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--run_name", default="lng_knp")
    ap.add_argument("--max_length", type=int, default=64)
    ap.add_argument("--per_device_train_batch_size", type=int, default=96)
    ap.add_argument("--learning_rate", type=float, default=1e-4)
    ap.add_argument("--warmup_ratio", type=float, default=0.04)
    ap.add_argument("--epochs", type=float, default=3)
    ap.add_argument("--max_steps", type=int, default=-1)
    ap.add_argument("--logging_steps", type=int, default=50)
    ap.add_argument("--save_steps", type=int, default=15_000)
    ap.add_argument("--resume_from_checkpoint", default=None,
                    help="an output_dir/checkpoint-<step> directory: weights, AdamW moments, step count and RNG states are restored "
                         "and the loop continues at the recorded step (HF TrainingArguments.resume_from_checkpoint)")
    ap.add_argument("--task_names", default=None, help='JSON list, e.g. ["rank","rank_4"]')
    ap.add_argument("--ln_to_weight", default=None, help="JSON dict of task weights (only 1.0 is built)")
    ap.add_argument("--use_fp16", action="store_true", help="bf16 GEMM operands, like the reference (main.py:152 bf16=args.use_fp16)")
    ap.add_argument("--wandb_project_name", default=None, help="accepted and ignored (no network)")
    ap.add_argument("--local_rank", type=int, default=-1)
    return ap.parse_args(argv)


def main(argv=None):
    import torch.distributed as dist
    from .dataset.lng_knp import LngKnpMarginMSEforT5SeqAQCollator, LngKnpMarginMSEforT5SeqAQDataset
    from .modeling.t5_generative_retriever import T5SeqAQEncoderForLngKnpMarginMSE
    from .tasks.trainer import LngKnpTrainer, LngKnpTrainingArgs
    args = get_args(argv)
    if args.loss_type != "t5seq_aq_encoder_lng_knp_margin_mse" or args.model_type != "t5_docid_gen_encoder":
        raise NotImplementedError(f"loss_type {args.loss_type!r} is outside this repository's path (SURVEY.md §2); "
                                  "built: t5seq_aq_encoder_lng_knp_margin_mse")
    local_rank = max(0, int(args.local_rank if args.local_rank >= 0 else os.environ.get("LOCAL_RANK", 0)))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=os.environ.get("RPR_DIST_BACKEND", "nccl"))
    dataset = LngKnpMarginMSEforT5SeqAQDataset(dataset_path=args.teacher_score_path, document_dir=args.collection_path,
                                               query_dir=args.queries_path, docid_to_smtid_path=args.docid_to_smtid_path,
                                               smtid_as_docid=args.smtid_as_docid)
    collator = LngKnpMarginMSEforT5SeqAQCollator(args.model_name_or_path, max_length=args.max_length)
    model = T5SeqAQEncoderForLngKnpMarginMSE.from_pretrained(args.resume_from_checkpoint or args.pretrained_path)
    model.to(local_rank)
    targs = LngKnpTrainingArgs(output_dir=args.output_dir, learning_rate=args.learning_rate, warmup_ratio=args.warmup_ratio,
                               per_device_train_batch_size=args.per_device_train_batch_size, num_train_epochs=args.epochs,
                               max_steps=args.max_steps, logging_steps=args.logging_steps, save_steps=args.save_steps,
                               bf16=args.use_fp16, task_names=json.loads(args.task_names) if args.task_names else None,
                               ln_to_weight=json.loads(args.ln_to_weight) if args.ln_to_weight else {})
    os.makedirs(args.output_dir, exist_ok=True)
    trainer = LngKnpTrainer(model, dataset, collator, targs)
    if trainer.rank == 0:
        print(f"lng_knp fine-tune: {len(dataset)} examples, {trainer.world} rank(s) x {targs.per_device_train_batch_size}, "
              f"{trainer.max_steps} steps ({trainer.warmup_steps} warm-up), lr {targs.learning_rate}, "
              f"{'bf16' if targs.bf16 else 'fp32-equivalent'} GEMMs")
    trainer.train(resume_from_checkpoint=args.resume_from_checkpoint or None)
    trainer.save_torch_model_and_tokenizer(collator.tokenizer)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
