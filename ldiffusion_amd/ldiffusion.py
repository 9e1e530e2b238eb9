"""Host-side mirror of the reference's orchestrator `LDiffusionModel` (/root/reference/ldiffusion.py:31-324) for the
sampling path: same constructor and `inference(...)` signature (cell and tissue levels), same error for an invalid level.
Training (`train`, `train_ldiffusion`: DeepSpeed ZeRO-3 fine-tuning, ldiffusion.py:121-315): the arithmetic core of a step lives in
`ldiffusion_amd.train` (forward and backward on the HIP kernels, parity-tested against torch.autograd over the oracle); the
orchestration around it (dataset, VGG19 content loss, ZeRO-3) is not built and `train` raises.
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

    def train(self, args, component="all", ldiffusion_weight=None, **_ignored):
        raise NotImplementedError("LDiffusionModel.train (ldiffusion.py:121-315) is not wired end to end: its data loading, the VGG19 content "
                                  "loss and DeepSpeed ZeRO-3 are outside this build.  The arithmetic of the step -- V5 feature loop, contrastive "
                                  "loss, backward through the VAE decoder and the UNet on the HIP kernels, gradient all-reduce, AdamW -- is "
                                  "ldiffusion_amd.train.train_step (tests/test_gpu_train.py)")

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
