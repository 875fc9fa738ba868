"""The optimisation loop around the prefix-oriented ranking fine-tune step: what the reference gets
from HF ``Trainer`` + ``CondDocID_DRTrainer`` (tasks/trainer.py:203-275, main.py:127-186) for the
``t5seq_aq_encoder_lng_knp_margin_mse`` loss, restated for a model whose forward, backward and AdamW run in
libripor_hip.so (``T5SeqAQEncoderForLngKnpMarginMSE.training_step``):

* data-parallel sampling: ``DistributedSampler(shuffle=True, seed)`` re-seeded per epoch, ``per_device_train_batch_size``
  examples per rank and step, incomplete last batches kept (HF default ``dataloader_drop_last=False``);
* learning rate: linear warm-up over ``ceil(max_steps * warmup_ratio)`` steps to ``learning_rate``, then linear decay to 0 at
  ``max_steps`` — HF ``get_linear_schedule_with_warmup`` as ``TrainingArguments(warmup_ratio=…, lr_scheduler_type=
  "linear")`` instantiates it (main.py:135-137; full_lng_knp_train_pipline.sh:80-99: lr 1e-4, warmup_ratio 0.04);
* every step: loss = sum of the task losses (``ln_to_weight`` 1 each), gradient all-reduce across the ranks overlapped with
  the backward (RCCL, engine.GradExchange), ``clip_grad_norm_(1.0)``, AdamW(betas (0.9, 0.999), eps 1e-8, weight_decay 0);
* checkpoints every ``save_steps`` under ``output_dir/checkpoint-<step>`` (at most ``save_total_limit``), resumable like HF
  ``Trainer`` checkpoints (``optimizer.pt``: the AdamW moments and step count, ``rng_state.pth``, ``trainer_state.json``;
  ``train(resume_from_checkpoint=…)`` continues at the recorded step of the recorded epoch's shuffle), and the final
  model under ``output_dir/checkpoint`` in the layout ``T5SeqAQEncoder.from_pretrained`` reads (config.json +
  pytorch_model.bin under the reference's tensor names + tokenizer files), like ``save_torch_model_and_tokenizer``.
"""
from __future__ import annotations

import json
import math
import os
import shutil
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import torch


def get_warmup_steps(max_steps: int, warmup_ratio: float, warmup_steps: int = 0) -> int:
    """HF ``TrainingArguments.get_warmup_steps``."""
    return warmup_steps if warmup_steps > 0 else math.ceil(max_steps * warmup_ratio)


def linear_schedule_with_warmup(step: int, num_warmup_steps: int, num_training_steps: int) -> float:
    """Multiplier of the base learning rate at optimizer step ``step`` (0-based): HF ``get_linear_schedule_with_warmup``."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))


@dataclass
class LngKnpTrainingArgs:
    """The fields of the reference's ``CondDocID_TrainingArgs`` this loop uses (main.py:131-155)."""
    output_dir: str
    learning_rate: float = 1e-4
    warmup_ratio: float = 0.04
    per_device_train_batch_size: int = 96
    num_train_epochs: float = 3
    max_steps: int = -1
    logging_steps: int = 50
    save_steps: int = 15_000
    save_total_limit: int = 5
    seed: int = 2
    max_grad_norm: float = 1.0
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    bf16: bool = False              # --use_fp16 in the reference's scripts means bf16 autocast (main.py:152); off unless asked, like
                                    # TrainingArguments.bf16 and main.py's flag
    task_names: Optional[List[str]] = None
    ln_to_weight: Dict[str, float] = field(default_factory=dict)


class LngKnpTrainer:
    def __init__(self, model, train_dataset, data_collator, args: LngKnpTrainingArgs, log: Callable[[str], None] = print):
        import torch.distributed as dist
        self.model, self.dataset, self.collator, self.args, self.log = model, train_dataset, data_collator, args, log
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        for name, w in (args.ln_to_weight or {}).items():
            if float(w) != 1.0:   # the device-side backward sums the task losses with unit weights (arguments.py:109-119 default)
                raise NotImplementedError(f"ln_to_weight[{name!r}] = {w}: only unit task weights are built")
        per_epoch = math.ceil(math.ceil(len(train_dataset) / self.world) / args.per_device_train_batch_size)
        self.steps_per_epoch = max(1, per_epoch)
        self.max_steps = args.max_steps if args.max_steps > 0 else math.ceil(args.num_train_epochs * self.steps_per_epoch)
        self.warmup_steps = get_warmup_steps(self.max_steps, args.warmup_ratio)
        self.global_step = 0
        self.history: List[dict] = []

    def lr_at(self, step: int) -> float:
        return self.args.learning_rate * linear_schedule_with_warmup(step, self.warmup_steps, self.max_steps)

    def _loader(self, epoch: int, skip_batches: int = 0):
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(self.dataset, num_replicas=self.world, rank=self.rank, shuffle=True, seed=self.args.seed)
        sampler.set_epoch(epoch)
        # resume: the batches of this epoch that were consumed before the checkpoint are dropped from the INDEX list, so
        # no item is fetched for them (the dataset draws negatives from Python's RNG, whose state the checkpoint restored)
        indices = list(sampler)[skip_batches * self.args.per_device_train_batch_size:]
        return DataLoader(self.dataset, batch_size=self.args.per_device_train_batch_size, sampler=indices,
                          collate_fn=self.collator, drop_last=False)

    def train(self, resume_from_checkpoint: Optional[str] = None):
        import random
        a = self.args
        random.seed(a.seed)          # the dataset draws its negatives from Python's RNG (dataset.py:479-483)
        torch.manual_seed(a.seed)
        ctx = self.model.base_model.engine_model().ctx
        saved_prec = ctx.get_precision() if hasattr(ctx, "get_precision") else None
        skip = 0
        if resume_from_checkpoint:
            skip = self.load_checkpoint_state(resume_from_checkpoint)
        try:
            if a.bf16:               # the reference's autocast: every GEMM operand of the step rounded to bf16, fp32 accumulation
                ctx.set_precision("bf16")
            t0, epoch, window = time.time(), self.global_step // self.steps_per_epoch, []
            while self.global_step < self.max_steps:
                for batch in self._loader(epoch, skip):
                    if self.global_step >= self.max_steps:
                        break
                    lr = self.lr_at(self.global_step)
                    losses = self.model.training_step(lr=lr, max_grad_norm=a.max_grad_norm, betas=(a.adam_beta1, a.adam_beta2),
                                                      eps=a.adam_epsilon, weight_decay=a.weight_decay, **batch)
                    self.global_step += 1
                    window.append(losses)
                    if self.global_step % a.logging_steps == 0 or self.global_step == self.max_steps:
                        mean = {k: float(torch.stack([w[k] for w in window]).mean()) for k in window[0]}   # one sync per log line
                        rec = dict(step=self.global_step, epoch=self.global_step / self.steps_per_epoch, learning_rate=lr,
                                   loss=sum(mean.values()), **mean, elapsed_s=time.time() - t0)
                        self.history.append(rec)
                        if self.rank == 0:
                            self.log(json.dumps(rec))
                        window = []
                    if a.save_steps > 0 and self.global_step % a.save_steps == 0:
                        ck = os.path.join(a.output_dir, f"checkpoint-{self.global_step}")
                        if self.rank == 0:
                            self.save_checkpoint(ck)          # weights, optimizer state, rank 0's RNG state
                        if self.world > 1:
                            import torch.distributed as dist
                            dist.barrier()                    # the directory exists
                            if self.rank != 0:
                                self.save_rng_state(ck)       # every rank draws its own negatives: one RNG file per rank, like HF
                            dist.barrier()
                        if self.rank == 0:
                            self._rotate()
                epoch += 1
                skip = 0
        finally:
            # the ctx is shared with the search / evaluation paths of the same process: a training run must not leave them
            # on bf16 GEMM operands (also when a step raises)
            if a.bf16 and saved_prec is not None:
                ctx.set_precision(saved_prec)
        return self.history

    # ---- checkpoints ------------------------------------------------------------------------------------------------
    def save_checkpoint(self, path: str, tokenizer=None):
        """config.json + pytorch_model.bin under the reference checkpoint's tensor names (DeviceModel.export_state_dict:
        the weights as the optimizer left them on the device) [+ tokenizer files]."""
        from ..modeling.t5_generative_retriever import expected_keys
        os.makedirs(path, exist_ok=True)
        base = self.model.base_model
        sd = {k: v.cpu() for k, v in base.engine_model().export_state_dict().items()}
        base._sd = {k: sd[k] for k in expected_keys(base.config)}   # host copy follows the device (the binding stays)
        base.save_pretrained(path)
        if tokenizer is not None:
            tokenizer.save_pretrained(path)
        with open(os.path.join(path, "trainer_state.json"), "w") as f:
            json.dump(dict(global_step=self.global_step, max_steps=self.max_steps, warmup_steps=self.warmup_steps,
                           learning_rate=self.args.learning_rate, steps_per_epoch=self.steps_per_epoch, world=self.world,
                           log_history=self.history), f, indent=1)
        # what HF Trainer's optimizer.pt / rng_state.pth carry (tasks/trainer.py of the reference inherits them): the AdamW
        # moments and step count of the flat device buffers, and the RNG states the data order and the negatives depend on
        import random
        st = self.model.train_state() if hasattr(self.model, "train_state") else None
        if st is not None:
            torch.save(dict(exp_avg=st.exp_avg.cpu(), exp_avg_sq=st.exp_avg_sq.cpu(), step=int(st.step), total=int(st.total)),
                       os.path.join(path, "optimizer.pt"))
        self.save_rng_state(path)

    def _rng_file(self, path: str) -> str:
        """HF Trainer's naming: rng_state.pth for one process, rng_state_<rank>.pth per process of a distributed run."""
        return os.path.join(path, "rng_state.pth" if self.world == 1 else f"rng_state_{self.rank}.pth")

    def save_rng_state(self, path: str):
        """This rank's Python / torch RNG states (the dataset draws its negatives with random.sample: every rank consumes its
        own stream, so a multi-rank run resumes as the same run only if every rank gets ITS state back)."""
        import random
        torch.save(dict(python=random.getstate(), torch=torch.get_rng_state()), self._rng_file(path))

    def load_checkpoint_state(self, path: str) -> int:
        """Restore what ``save_checkpoint`` wrote beside the weights (the caller loads those: ``from_pretrained(path)``):
        global step, log history, AdamW moments, RNG states. Returns the number of batches of the current epoch to skip."""
        import random
        with open(os.path.join(path, "trainer_state.json")) as f:
            ts = json.load(f)
        if ts.get("steps_per_epoch", self.steps_per_epoch) != self.steps_per_epoch or ts.get("world", self.world) != self.world:
            raise ValueError(f"{path}: written with {ts.get('world')} rank(s) x {ts.get('steps_per_epoch')} steps per epoch, "
                             f"this run has {self.world} x {self.steps_per_epoch}: the data order would not continue")
        self.global_step, self.history = int(ts["global_step"]), list(ts.get("log_history", []))
        opt = os.path.join(path, "optimizer.pt")
        if os.path.exists(opt) and hasattr(self.model, "train_state"):
            o = torch.load(opt, map_location="cpu")
            st = self.model.train_state()
            if int(o["total"]) != int(st.total):
                raise ValueError(f"{opt}: {o['total']} parameters, the model has {st.total}")
            st.exp_avg.copy_(o["exp_avg"]); st.exp_avg_sq.copy_(o["exp_avg_sq"]); st.step = int(o["step"])
        rng = self._rng_file(path)
        if not os.path.exists(rng) and self.world > 1:
            raise ValueError(f"{path}: no RNG state of rank {self.rank} ({os.path.basename(rng)}): the checkpoint was not written by "
                             f"a {self.world}-rank run of this trainer")
        if os.path.exists(rng):
            r = torch.load(rng, map_location="cpu", weights_only=False)
            random.setstate(r["python"]); torch.set_rng_state(r["torch"])
        return self.global_step % self.steps_per_epoch

    def _rotate(self):
        lim = self.args.save_total_limit
        if not lim or lim <= 0:
            return
        cks = sorted((int(d.split("-")[1]), d) for d in os.listdir(self.args.output_dir)
                     if d.startswith("checkpoint-") and d.split("-")[1].isdigit())
        for _, d in cks[:-lim]:
            shutil.rmtree(os.path.join(self.args.output_dir, d), ignore_errors=True)

    def save_torch_model_and_tokenizer(self, tokenizer=None):
        """reference tasks/trainer.py: the final model under ``output_dir/checkpoint``."""
        if self.rank == 0:
            self.save_checkpoint(os.path.join(self.args.output_dir, "checkpoint"), tokenizer)
