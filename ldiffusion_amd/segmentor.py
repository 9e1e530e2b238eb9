"""Host-side mirror of the reference's `Segmentor` for the sampling path (same names, argument meaning and errors).

Mirrors /root/reference/segmentor.py: `_resolve_ldiffusion_dir` (:26-29), `_ensure_ldiffusion_proj` (:31-52),
`_get_text_embeddings` (:54-60), `load_ldiffusion` (:76-84), `ldiffusion_augment` (:86-112), `_UNetTextAlignWrapper`
(:183-205), the sampler part + mask tail of `inference_cell_model` (:490-545) and of `inference_tissue_model_nnUNetv2`
(:388-488).  The segmentation heads themselves (`model/conductor.py`, nnU-Net) are outside the hot-path scope (SURVEY.md
8a/8f): both inference functions take the head as a callable and raise if none is given instead of silently substituting one;
for the tissue level the callable is applied through the device-resident mirror of nnU-Net's sliding-window predictor
(`tiling.predict_sliding_window_return_logits`: Gaussian-weighted fp16 accumulation, mirroring TTA) instead of the reference's
PNG / tmpdir / multiprocess round trip.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .models import UNet2DConditionModel
from .pipeline import PROMPT, LaplaceSampler, StableDiffusionImg2ImgPipeline, argmax_mask

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # segmentor.py:508
IMAGENET_STD = (0.229, 0.224, 0.225)


class TextAlignedUNet:
    """`_UNetTextAlignWrapper` (segmentor.py:183-205): when `encoder_hidden_states` is None or its last dim differs from
    `cross_attention_dim`, the cached default embeddings are used instead, expanded to the batch."""

    def __init__(self, base_unet, default_text_embeddings):
        self.base_unet = base_unet
        self.default_text_embeddings = default_text_embeddings
        self.cross_attention_dim = base_unet.config.cross_attention_dim
        self.config = base_unet.config

    def __call__(self, sample, timestep, encoder_hidden_states, *args, **kwargs):
        use_fallback = encoder_hidden_states is None or encoder_hidden_states.shape[-1] != self.cross_attention_dim
        if use_fallback:
            emb = self.default_text_embeddings
            if emb.shape[0] != sample.shape[0]:
                emb = emb[:1].expand(sample.shape[0], -1, -1)
            encoder_hidden_states = emb
        return self.base_unet(sample, timestep, encoder_hidden_states.to(sample.device, dtype=torch.float32), *args, **kwargs)

    forward = __call__

    def eval(self):
        return self

    def to(self, *a, **k):
        return self


class Segmentor:
    def __init__(self, train_loader, val_loader, level, num_classes):
        if not torch.cuda.is_available():
            raise RuntimeError("ldiffusion_amd.Segmentor needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device(f"cuda:{torch.cuda.current_device()}")
        self.level = level
        self.num_classes = num_classes
        self.model = None
        self.train_loader, self.val_loader = train_loader, val_loader
        self.ldiffusion_proj = None
        self._sampler = None

    def _resolve_ldiffusion_dir(self, ldiffusion_weight):
        return ldiffusion_weight if os.path.isdir(ldiffusion_weight) else os.path.dirname(ldiffusion_weight)

    def _ensure_ldiffusion_proj(self, pipeline, unet, ldiffusion_weight=None):
        hidden = pipeline.text_encoder.config.hidden_size
        cad = unet.config.cross_attention_dim
        if self.ldiffusion_proj is None or self.ldiffusion_proj.in_features != hidden or self.ldiffusion_proj.out_features != cad:
            self.ldiffusion_proj = nn.Linear(hidden, cad).to(self.device, dtype=torch.float32)
        if ldiffusion_weight is not None:
            p = os.path.join(self._resolve_ldiffusion_dir(ldiffusion_weight), "proj_weights.pt")
            if os.path.exists(p):
                self.ldiffusion_proj.load_state_dict(torch.load(p, map_location=self.device), strict=True)
        self.ldiffusion_proj = self.ldiffusion_proj.to(self.device, dtype=torch.float32).eval()
        return self.ldiffusion_proj

    @torch.no_grad()
    def _get_text_embeddings(self, prompt, batch_size, pipeline, unet):
        proj = self._ensure_ldiffusion_proj(pipeline, unet)
        ids = torch.tensor(pipeline.tokenizer([prompt] * batch_size)["input_ids"], dtype=torch.long, device=self.device)
        emb = pipeline.text_encoder(ids)["last_hidden_state"].to(dtype=torch.float32)
        return proj(emb).to(dtype=torch.float32)

    def initialize_model(self, level, num_classes):
        if level not in ("tissue", "cell"):
            raise ValueError("Invalid level specified. Choose 'tissue' or 'cell'.")
        raise RuntimeError("the segmentation heads (model/conductor.py, nnU-Net) are outside the MI355X hot-path scope; pass `head=`")

    def load_ldiffusion(self, ldiffusion_weight, diffusion_path):
        pipeline = StableDiffusionImg2ImgPipeline.from_pretrained(diffusion_path, torch_dtype=torch.float32, device=self.device)
        unet = UNet2DConditionModel.from_pretrained(ldiffusion_weight, device=self.device).eval()
        vae = pipeline.vae
        pipeline.unet = unet
        if pipeline.text_encoder is not None:
            self._ensure_ldiffusion_proj(pipeline, unet, ldiffusion_weight=ldiffusion_weight)
        return pipeline, unet, vae

    def _one_pass(self, images, text_embeddings, pipeline, unet):
        if self._sampler is None or self._sampler.pipeline is not pipeline or pipeline.unet is not unet:
            pipeline.unet = unet
            self._sampler = LaplaceSampler(pipeline)
        return self._sampler.sample(images, text_embeddings, 1, want_features=False, want_rgb=True)

    def build_augmented_dataloader(self, dataloader, augment_fn, pipeline, unet, vae, device, batch_size, category, num_workers=0):
        """segmentor.py:144-161: run `augment_fn(inputs, pipeline, unet, vae)` (e.g. `ldiffusion_augment`) once over `dataloader` -- batches
        (inputs, masks, _) -- under no_grad, keep the augmented inputs and the masks on the host, and serve them from a shuffling
        TensorDataset loader of `batch_size`.  `category` only labels the reference's progress bar."""
        from torch.utils.data import DataLoader, TensorDataset
        all_aug_inputs, all_masks = [], []
        for inputs, masks, _ in dataloader:
            inputs, masks = inputs.to(device), masks.to(device)
            with torch.no_grad():
                aug_inputs = augment_fn(inputs, pipeline, unet, vae)
            all_aug_inputs.append(aug_inputs.cpu())
            all_masks.append(masks.cpu())
        dataset = TensorDataset(torch.cat(all_aug_inputs, dim=0), torch.cat(all_masks, dim=0))
        return DataLoader(dataset, batch_size=batch_size, shuffle=True, num_workers=num_workers)

    @torch.no_grad()
    def ldiffusion_augment(self, inputs, pipeline, unet, vae, text_embeddings=None):
        """segmentor.py:86-112, batched: encode mean -> set_timesteps(1) -> one UNet pass -> step -> decode -> uint8 ->
        Resize((1024,1024)) + ToTensor.  Returns float [B,3,1024,1024] in [0,1] on the device."""
        if text_embeddings is None:
            text_embeddings = self._get_text_embeddings(PROMPT, 1, pipeline, unet)
        out = self._one_pass(inputs.to(self.device, dtype=torch.float32), text_embeddings, pipeline, unet)
        rgb = out["rgb"]  # [B,H,W,3] u8
        if rgb.shape[1:3] == (1024, 1024):
            return rgb.permute(0, 3, 1, 2).float() / 255.0
        from PIL import Image  # other sizes: the reference's PIL bilinear Resize, on the host
        ims = [np.asarray(Image.fromarray(a).resize((1024, 1024), Image.BILINEAR), np.float32) / 255.0 for a in rgb.cpu().numpy()]
        return torch.from_numpy(np.stack(ims)).permute(0, 3, 1, 2).contiguous().to(self.device)

    @torch.no_grad()
    def ldiffusion_augment_for_multimodal(self, rgb, dtm, pipeline, unet, vae, controlnet, batch_size, device, u=None, seed=0):
        """segmentor.py:301-386 (sampler variant V7: RGB + depth, ControlNet residuals), per sample as there:
            rgb, dtm -> bilinear 256 x 256;  z = vae.encode(rgb).latent_dist.sample() * 0.18215          (:339)
            depth = bilinear 32 x 32 of dtm, repeated over the 4 latent channels                          (:340-341)
            z_noisy = z + Laplace(0, 1) * depth                                                           (:344-345)
            ctx = proj(text_encoder("A remote sense image", padded to 77))                                (:348-352)
            residuals = controlnet(z_noisy, t, ctx, controlnet_cond = depth x 3)  at set_timesteps(1)     (:355-363)
            eps = unet(z_noisy, t, ctx, down_block_additional_residuals, mid_block_additional_residual)   (:366-372)
            recon = vae.decode((z_noisy - eps * depth) / 0.18215).sample                                  (:375-379)
        Returns the list of [256, 256, 3] float arrays the reference returns (:381-386).  The ControlNet is the caller's module (the reference
        trains none and ships no weights for one: SURVEY 8f); the UNet, the VAE and the Laplace transform run on the HIP kernels.
        `u` (optional, [B, 4, 32, 32] uniform draws in (-1, 1)) fixes the Laplace noise (parity is defined given u)."""
        from .pipeline import laplace_noise
        dev = self.device
        proj = self._ensure_ldiffusion_proj(pipeline, unet)
        rgb = F.interpolate(rgb.to(dev, torch.float32), size=(256, 256), mode="bilinear", align_corners=False)
        dtm = F.interpolate(dtm.to(dev, torch.float32), size=(256, 256), mode="bilinear", align_corners=False)
        ids = pipeline.tokenizer(["A remote sense image"], padding="max_length", max_length=77, return_tensors="pt")
        ids = torch.as_tensor(ids["input_ids"] if isinstance(ids, dict) else ids.input_ids, dtype=torch.long, device=dev)
        ctx = proj(pipeline.text_encoder(ids)["last_hidden_state"].to(device=dev, dtype=torch.float32)).to(dtype=torch.float32)
        recon = []
        for i in range(dtm.shape[0]):
            dtm_i, rgb_i = dtm[i], rgb[i].unsqueeze(0)
            depth_condition = dtm_i.unsqueeze(0).repeat(1, 3, 1, 1)
            latents = vae.encode(rgb_i).latent_dist.sample().to(torch.float32) * 0.18215
            depth = F.interpolate(dtm_i.unsqueeze(0), size=(32, 32), mode="bilinear", align_corners=False).repeat(1, latents.shape[1], 1, 1)
            # z + Laplace(0, 1) * depth: the device transform with unit scale gives z + n, so noise = that minus z
            noise = laplace_noise(torch.zeros_like(latents), 1.0, u=None if u is None else u[i:i + 1].to(dev), seed=seed, offset=i * latents.numel())
            noisy = latents + noise * depth
            pipeline.scheduler.set_timesteps(1, device=dev)
            for t in pipeline.scheduler.timesteps:
                down, mid = controlnet(sample=noisy, timestep=t, encoder_hidden_states=ctx, controlnet_cond=depth_condition, return_dict=False)
                eps = unet(noisy, t, encoder_hidden_states=ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
            den = noisy - eps * depth
            img = vae.decode(den / 0.18215).sample
            recon.append(img.squeeze(0).permute(1, 2, 0).cpu().numpy())
        return recon

    @torch.no_grad()
    def inference_cell_model(self, image_path, diffusion_path, ldiffusion_weight, segmentor_weight, head=None, text_embeddings=None):
        """segmentor.py:490-545 with the head injected: `head(decoded_rgb_float[1,3,1024,1024] normalised) -> logits [1,C,H,W]`."""
        from PIL import Image
        if head is None:
            raise RuntimeError("inference_cell_model: CellSegClassifier (Cellpose + ResNet152) is outside the hot-path scope; pass `head=`")
        pipeline, unet, _ = self.load_ldiffusion(ldiffusion_weight, diffusion_path)
        image = Image.open(image_path).convert("RGB")
        width, height = image.size
        mean = torch.tensor(IMAGENET_MEAN, device=self.device).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_STD, device=self.device).view(1, 3, 1, 1)
        x = torch.from_numpy(np.asarray(image.resize((1024, 1024), Image.BILINEAR), np.float32) / 255.0).permute(2, 0, 1)[None].to(self.device)
        x = (x - mean) / std
        if text_embeddings is None:
            text_embeddings = self._get_text_embeddings(PROMPT, 1, pipeline, unet)
        out = self._one_pass(x, text_embeddings, pipeline, unet)
        decoded = Image.fromarray(out["rgb"][0].cpu().numpy())
        self._sampler.check_finite()   # fp16 overflow in either graph raises here instead of yielding a plausible-looking mask
        model_input = (out["rgb"].permute(0, 3, 1, 2).float() / 255.0 - mean) / std
        logits = head(model_input)
        mask = argmax_mask(logits)[0].cpu().numpy()
        mask = np.array(Image.fromarray(mask.astype(np.uint8)).resize((width, height), resample=Image.NEAREST))
        return decoded.resize((width, height), Image.BILINEAR), mask

    @torch.no_grad()
    def inference_tissue_model_nnUNetv2(self, image_path, diffusion_path, ldiffusion_weight, segmentor_weight, output_path=None,
                                        predictor=None, text_embeddings=None, tile_size=(512, 512), tile_step_size=0.5,
                                        mirror_axes=(0, 1), num_heads=None):
        """segmentor.py:388-488 with the tissue head injected.
        * `image_path` a directory: the reference hands the folder to nnU-Net's file predictor and returns `(None, None)`
          (:399-421).  Here `predictor(image_path, output_path)` is called if it accepts two arguments (a file-level predictor),
          else every image of the folder goes through the single-image path and its mask is written as PNG to `output_path`.
        * a single image: Resize(1024) + ImageNet-normalise -> one-pass sampler ONLY when width == height (:427); other aspect
          ratios skip the diffusion and feed the image as it is (:449-450).  The decoded RGB stays on the device, is handed to
          `predictor([1, 3, th, tw] float in 0..255 scale as the PNG would hold) -> [1, heads, th, tw]` tile by tile
          (nnU-Net order, Gaussian fp16 accumulation, mirroring), arg-max on the device, and `(decoded PIL image, uint8 mask)`
          is returned like :486-488."""
        from PIL import Image
        from . import tiling
        if predictor is None:
            raise RuntimeError("inference_tissue_model_nnUNetv2: the nnU-Net tissue head is outside the hot-path scope; pass `predictor=`")
        if os.path.isdir(image_path):
            if not output_path:
                raise ValueError("When image_path is a folder, output_path must be specified!")
            import inspect
            try:
                n_args = len(inspect.signature(predictor).parameters)
            except (TypeError, ValueError):
                n_args = 1
            if n_args >= 2:
                predictor(image_path, output_path)
                return None, None
            os.makedirs(output_path, exist_ok=True)
            for name in sorted(os.listdir(image_path)):
                if name.lower().endswith((".png", ".jpg", ".jpeg", ".tif", ".tiff", ".bmp")):
                    _, m = self.inference_tissue_model_nnUNetv2(os.path.join(image_path, name), diffusion_path, ldiffusion_weight, segmentor_weight,
                                                                None, predictor, text_embeddings, tile_size, tile_step_size, mirror_axes, num_heads)
                    Image.fromarray(m).save(os.path.join(output_path, os.path.splitext(name)[0] + ".png"))
            return None, None   # batch mode returns no single mask (segmentor.py:421)
        pipeline, unet, _ = self.load_ldiffusion(ldiffusion_weight, diffusion_path)
        image = Image.open(image_path).convert("RGB")
        width, height = image.size
        if width == height:
            mean = torch.tensor(IMAGENET_MEAN, device=self.device).view(1, 3, 1, 1)
            std = torch.tensor(IMAGENET_STD, device=self.device).view(1, 3, 1, 1)
            x = torch.from_numpy(np.asarray(image.resize((1024, 1024), Image.BILINEAR), np.float32) / 255.0).permute(2, 0, 1)[None].to(self.device)
            if text_embeddings is None:
                text_embeddings = self._get_text_embeddings(PROMPT, 1, pipeline, unet)
            rgb = self._one_pass((x - mean) / std, text_embeddings, pipeline, unet)["rgb"]   # [1,1024,1024,3] u8, on the device
            decoded = Image.fromarray(rgb[0].cpu().numpy())
            self._sampler.check_finite()   # fp16 overflow in either graph raises here instead of yielding a plausible-looking mask
        else:                                                                                 # non-square: no diffusion (segmentor.py:449-450)
            rgb = torch.from_numpy(np.array(image, np.uint8))[None].to(self.device)
            decoded = image
        data = rgb[0].permute(2, 0, 1).float()                                                # what the PNG the reference writes would hold
        heads = int(num_heads if num_heads is not None else self.num_classes)
        th, tw = min(tile_size[0], data.shape[1]), min(tile_size[1], data.shape[2])
        logits = tiling.predict_sliding_window_return_logits(data, predictor, heads, (th, tw), tile_step_size, True, mirror_axes)
        mask = argmax_mask(logits[None].float())[0].cpu().numpy()
        return decoded, mask
