"""Host-side mirror of the reference's orchestrator `LDiffusionModel` (/root/reference/ldiffusion.py:31-324) for the
sampling path: same constructor and `inference(...)` signature (cell and tissue levels), same error for an invalid level.
Training: `train_ldiffusion` (ldiffusion.py:121-295) runs on the HIP kernels, forward and backward (`ldiffusion_amd.train`,
parity-tested against torch.autograd over the oracle), with the data loader injected; what differs from the reference (no ZeRO-3, optional
VGG19 content term, fp16 operands) is listed in its docstring.  The segmentor stage of `train` trains the out-of-scope heads and raises.
"""
from __future__ import annotations

import os

import torch

from .parallel import world_info
from .segmentor import Segmentor


class LDiffusionModel:
    def __init__(self, diffusion_path, level, local_rank=-1):
        rank, world, env_local = world_info()                     # ldiffusion.py:34-35,42
        self.world_size, self.rank = world, rank
        self.is_distributed = world > 1
        self.local_rank = int(local_rank if local_rank is not None and local_rank >= 0 else env_local)
        if not torch.cuda.is_available():
            raise RuntimeError("ldiffusion_amd.LDiffusionModel needs a ROCm GPU (the MI355X path has no CPU fallback)")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device(f"cuda:{self.local_rank}")
        self.diffusion_path = diffusion_path
        self.level = level
        self.linear_layer = None

    def _is_main_process(self):
        return self.rank == 0

    def load_model(self, model_path):
        """ldiffusion.py:66-70 -> (pipeline, vae)"""
        from .pipeline import StableDiffusionImg2ImgPipeline
        pipeline = StableDiffusionImg2ImgPipeline.from_pretrained(model_path, torch_dtype=torch.float32, device=self.device)
        return pipeline, pipeline.vae

    def _reduce_mean(self, value):
        """ldiffusion.py:54-64: mean of a python float over the ranks."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return value
        t = torch.tensor([value], device=self.device, dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item() / dist.get_world_size())

    def train_ldiffusion(self, args, train_loader, val_loader=None):
        """ldiffusion.py:121-295 on the HIP kernels (forward and backward, ldiffusion_amd.train): per batch Resize(64) -> VAE encode ->
        V5 loop -> InfoNCE loss -> backward -> gradient all-reduce -> clip 1.0 -> AdamW(lr 1e-5, wd 0.01) on UNet + text projection; per epoch
        the mean loss goes to train_save/loss/<yy_mm_dd>/contrast_loss.csv and, when it improved, the UNet (diffusers layout) and
        proj_weights.pt are saved under LDiffusion/train_save/unet/<yy_mm_dd>/ (the layout `Segmentor.load_ldiffusion` reads back).
        Differences, all explicit: replicated parameters + one gradient all-reduce instead of ZeRO-3 with CPU offload; fp16 MFMA operands
        with fp32 accumulation / gradients / master weights instead of fp32 everywhere; the VGG19 content term only when
        `args.vgg_features` (a callable) is given -- ImageNet weights are not available offline, and the loop says so once.
        `train_loader` yields (image [B,3,H,W] float in ToTensor range, _, label [B,1,H,W]) like the reference's PUMA loader."""
        import csv
        import time
        from datetime import datetime
        import torch.nn as nn
        import torch.nn.functional as F
        from . import train as T, weights
        from .loss import InfoNceLoss
        num_epochs = int(getattr(args, "ldiffusion_epochs", 10))                     # hard-coded 10 in the reference (:122)
        num_inference_steps = args.num_inference_steps
        root = getattr(args, "output_root", ".")
        self.pipeline, self.vae = self.load_model(args.diffusion_path)
        current_date = datetime.now().strftime("%y_%m_%d")
        csv_dir = os.path.join(root, "train_save", "loss", current_date)
        csv_file = os.path.join(csv_dir, "contrast_loss.csv")
        os.makedirs(csv_dir, exist_ok=True)
        if self._is_main_process():
            with open(csv_file, mode="w", newline="") as f:
                csv.writer(f).writerow(["epoch", "loss"])
        hidden = self.pipeline.text_encoder.config.hidden_size
        cad = self.pipeline.unet.config.cross_attention_dim
        if self.linear_layer is None or self.linear_layer.in_features != hidden or self.linear_layer.out_features != cad:
            self.linear_layer = nn.Linear(hidden, cad)
        self.linear_layer = self.linear_layer.to(self.device, dtype=torch.float32)
        unet = T.TrainableUNet(self.pipeline.unet._cfg, self.pipeline.unet.state_dict(), self.device)
        dec = T.FrozenVAEDecoder(self.vae._cfg, self.vae._host_sd, self.device)
        proj = (self.linear_layer.weight, self.linear_layer.bias)
        loss_obj = InfoNceLoss(vgg_features=getattr(args, "vgg_features", None))
        if loss_obj.vgg is None and self._is_main_process():
            print("[train_ldiffusion] no vgg_features given: the VGG19 content term of InfoNceLoss (model/loss.py:21-42) is left out; "
                  "training on the contrastive term only")
        save_path = os.path.join(root, "LDiffusion", "train_save", "unet", current_date)
        checkpoint = 100
        n_sched = min(int(num_inference_steps / 5), len(self.pipeline.scheduler.alphas_cumprod))
        state, noise_offset = {}, 0   # Philox counters consumed so far: every step draws fresh Laplace noise (the reference samples anew each time)
        # forward + loss + backward as ONE captured HIP graph per step (train.GraphedStep) where the loss is the contrastive term alone and the
        # batch has the captured shape; anything else (content term, a short last batch, more triples than the capture holds) runs eagerly
        use_graph = bool(getattr(args, "use_graph", True)) and loss_obj.vgg is None
        gstep = None
        # The reference's engine is DeepSpeed ZeRO stage 3 (ldiffusion.py:165-193): under a process group every rank owns 1/W of the float32 masters and
        # of the AdamW moments, gradients are reduce-scattered, the parameters all-gathered in front of each forward (train.ShardedAdamW,
        # partition_params).  Created BEFORE the GraphedStep: the parameters move into its flat buffer.  One process: the multi-tensor AdamW.
        opt = T.ShardedAdamW(unet.parameters() + list(proj), lr=1e-5, weight_decay=0.01, partition_params=True) if self.is_distributed else None
        for epoch in range(num_epochs):
            if hasattr(train_loader, "sampler") and hasattr(train_loader.sampler, "set_epoch"):
                train_loader.sampler.set_epoch(epoch)
            total, start = 0.0, time.time()
            for image, _, label in train_loader:
                B = image.shape[0]
                image = F.interpolate(image.to(self.device, dtype=torch.float32), size=(64, 64), mode="bilinear", align_corners=False, antialias=True)
                ids = torch.tensor(self.pipeline.tokenizer(["A pathological slide"] * B)["input_ids"], dtype=torch.long, device=self.device)
                with torch.no_grad():
                    text_hidden = self.pipeline.text_encoder(ids)["last_hidden_state"].to(dtype=torch.float32)
                    lab = F.interpolate(label.to(self.device, torch.float32), size=(64, 64), mode="bilinear", align_corners=False).to(torch.uint8)
                    latents = self.vae.encode(image).latent_dist.mean.to(dtype=torch.float32)
                self.pipeline.scheduler.set_timesteps(n_sched, device=self.device)
                ts = [int(t) for t in self.pipeline.scheduler.timesteps]

                def loss_fn(feats, rgb, image=image, lab=lab):
                    contrastive = loss_obj.compute_contrastive_loss(feats, lab)
                    if loss_obj.vgg is None:
                        return contrastive
                    big = F.interpolate(rgb, size=(1024, 1024), mode="bilinear", align_corners=False)
                    return loss_obj.compute_content_loss(image, big) + contrastive

                pairs = None
                if use_graph:
                    pairs = loss_obj.sample_triples(lab.to("cpu"))   # the draws compute_contrastive_loss would make (same torch random stream)
                    n_tr = sum(len(t) for t in pairs)
                    if gstep is None and n_tr > 0:
                        gstep = T.GraphedStep(unet, dec, proj, B, ts, self.pipeline.scheduler.alphas_cumprod, latent_hw=latents.shape[-1],
                                              text_len=text_hidden.shape[1], text_dim=text_hidden.shape[2], max_triples=max(256, 64 * B),
                                              num_negatives=loss_obj.num_negatives, temperature=loss_obj.temperature)
                    # never skipped per rank: a batch without sample triples still runs the (eager) step with zero gradients, so that every rank
                    # enters the same gradient collective (train.run_step)
                    val, _ = T.run_step(gstep, unet, dec, proj, latents, text_hidden, ts, self.pipeline.scheduler.alphas_cumprod, pairs, state,
                                        lr=1e-5, weight_decay=0.01, max_grad_norm=1.0, seed=self.rank, offset=noise_offset, optimizer=opt)
                    total += val
                else:
                    total += T.train_step(unet, dec, proj, latents, text_hidden, ts, self.pipeline.scheduler.alphas_cumprod, None, None, state, lr=1e-5,
                                          weight_decay=0.01, loss_fn=loss_fn, max_grad_norm=1.0, seed=self.rank, offset=noise_offset, optimizer=opt)
                noise_offset += len(ts) * latents.numel()   # v5_features draws offset + i * numel for pass i
            current = self._reduce_mean(total / max(1, len(train_loader)))
            if self._is_main_process():
                print(f"Epoch [{epoch + 1}/{num_epochs}], Loss: {current:.4f}, Elapsed Time: {time.time() - start}s")
            if current < checkpoint:
                if opt is not None:
                    opt.gather()   # (a collective: every rank) the full parameters, for the checkpoint rank 0 writes
                if self._is_main_process():
                    weights.save_model_dir(save_path, self.pipeline.unet._cfg, {k: p.detach().to("cpu") for k, p in unet.p.items()})
                    torch.save({"weight": proj[0].detach().to("cpu"), "bias": proj[1].detach().to("cpu")}, os.path.join(save_path, "proj_weights.pt"))
                checkpoint = current
            if self.is_distributed:
                torch.distributed.barrier()
            if self._is_main_process():
                with open(csv_file, mode="a", newline="") as f:
                    csv.writer(f).writerow([epoch + 1, current])
        del self.pipeline, self.vae
        torch.cuda.empty_cache()
        return save_path

    def train(self, args, component="all", ldiffusion_weight=None, train_loader=None, val_loader=None, **_ignored):
        """ldiffusion.py:297-315.  The PUMA dataset / dataloader (dataset.py, cv2) and the segmentation heads are outside this build: the
        loaders are injected, `component="ldiffusion"` runs the L-Diffusion warm-up on the HIP kernels and returns the saved weight
        directory; the segmentor stage raises."""
        if component not in ("all", "ldiffusion", "segmentor"):
            raise ValueError(f"unknown component {component!r}")
        if component in ("all", "ldiffusion"):
            if train_loader is None:
                raise RuntimeError("LDiffusionModel.train: the reference's dataset pipeline (load_data, dataset.py) is outside this build; pass train_loader=")
            if self._is_main_process():
                print("Starting LDiffusion warming up...")
            ldiffusion_weight = self.train_ldiffusion(args, train_loader, val_loader)
        if component in ("all", "segmentor"):
            raise NotImplementedError("segmentor training (train_cell_model / train_tissue_model_nnUNetv2, segmentor.py:163-299) trains the "
                                      f"out-of-scope heads; the L-Diffusion weights are at {ldiffusion_weight!r}")
        return ldiffusion_weight

    def inference(self, image_path, ldiffusion_weight, segmentor_weight, num_classes, head=None, predictor=None, output_path=None,
                  text_embeddings=None, **_readme_kwargs):
        """ldiffusion.py:317-324.  `head` (cell) / `predictor` (tissue) = the segmentation head callable (out of scope, see
        segmentor.py); `output_path` is the folder-mode argument of the tissue path; other README-era keyword arguments (dtm_path)
        are accepted and ignored like the code ignores them."""
        segmentor = Segmentor(train_loader=None, val_loader=None, level=self.level, num_classes=num_classes)
        if self.level == "tissue":
            return segmentor.inference_tissue_model_nnUNetv2(image_path, self.diffusion_path, ldiffusion_weight, segmentor_weight,
                                                             output_path=output_path, predictor=predictor if predictor is not None else head,
                                                             text_embeddings=text_embeddings)
        elif self.level == "cell":
            return segmentor.inference_cell_model(image_path, self.diffusion_path, ldiffusion_weight, segmentor_weight, head=head,
                                                  text_embeddings=text_embeddings)
        else:
            raise ValueError("Invalid level specified. Choose 'tissue' or 'cell'.")
