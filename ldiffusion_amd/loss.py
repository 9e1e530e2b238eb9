"""Host-side mirror of the reference's `InfoNceLoss` (/root/reference/model/loss.py:10-126) for the fine-tuning step: same method names,
argument meaning, hyper-parameters (temperature 0.5, 1024 negatives, 1 % of the positives as anchors) and the SAME consumption of the
torch random stream (`randperm` / `randint` on the host generator, in the same order), so that with one seed both draw the same
(anchor, positive, negatives) triples -- pinned to the reference's own code by tests/golden/reference_infonce.npz
(scripts/gen_golden_loss.py).

The VGG19 content term (`compute_content_loss`, model/loss.py:21-42) needs ImageNet weights the reference downloads at construction
(`vgg19(weights=VGG19_Weights.DEFAULT)`, :15); they are not available offline.  The term is therefore an injected callable
`vgg_features(x [B,3,224,224]) -> features`; without one `compute_loss` raises instead of silently dropping the term, and
`compute_contrastive_loss` / `sample_triples` are usable on their own.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class InfoNceLoss:
    def __init__(self, temperature=0.5, num_negatives=1024, eps=1e-8, vgg_features=None):
        self.temperature = temperature
        self.num_negatives = num_negatives
        self.eps = eps
        self.vgg = vgg_features

    def compute_content_loss(self, original_image, generated_image):
        """model/loss.py:21-42: both images bilinearly resized to 224 x 224, MSE between the VGG19 feature maps."""
        if self.vgg is None:
            raise RuntimeError("InfoNceLoss: the VGG19 content term needs ImageNet weights that are not available offline; pass vgg_features=")
        a = F.interpolate(original_image, size=(224, 224), mode="bilinear", align_corners=False)
        b = F.interpolate(generated_image, size=(224, 224), mode="bilinear", align_corners=False)
        return F.mse_loss(self.vgg(a), self.vgg(b))

    def sample_triples(self, labels):
        """The random draws of model/loss.py:62-87 for labels [B, 1, H, W] (or [B, H*W]): per image and per label value with more than one
        pixel and more than `num_negatives` pixels of other labels, max(1, 1 % of its pixels) anchors by `randperm`, for each one positive by
        `randint` and `num_negatives` negatives by `randperm`.  Returns, per image, a list of (anchor, positive, [negatives]) pixel indices."""
        B = labels.shape[0]
        labels = labels.reshape(B, -1)
        out = []
        for b in range(B):
            label = labels[b]
            triples = []
            for lbl in torch.unique(label):
                mask = label == lbl
                pos_idx = torch.nonzero(mask).squeeze(-1)
                neg_idx = torch.nonzero(~mask).squeeze(-1)
                if len(pos_idx) > 1 and len(neg_idx) > self.num_negatives:
                    sampled = torch.randperm(len(pos_idx))[:max(1, int(0.01 * len(pos_idx)))]
                    for idx in sampled:
                        anchor = pos_idx[idx].item()
                        pool = pos_idx[pos_idx != anchor]
                        if len(pool) == 0:
                            continue
                        positive = pool[torch.randint(0, len(pool), (1,))].item()
                        negatives = neg_idx[torch.randperm(len(neg_idx))[:self.num_negatives]].tolist()
                        triples.append((anchor, positive, negatives))
            out.append(triples)
        return out

    def compute_contrastive_loss(self, features, labels, triples=None):
        """model/loss.py:44-109: features [B, n, H, W], labels [B, 1, H, W]; mean over all triples of the cross-entropy of
        [anchor.positive | anchor.negatives] / temperature with the positive as target.  `triples` (from `sample_triples`) may be given."""
        B, n, H, W = features.shape
        feat = features.view(B, n, -1).permute(0, 2, 1)
        triples = self.sample_triples(labels.to("cpu") if triples is None and labels.is_cuda else labels) if triples is None else triples
        if features.is_cuda:
            # GPU features: value and gradient from the one-launch HIP kernel (ldiff_op_infonce); the loop below is the host statement of
            # the same arithmetic that tests/test_cpu_oracle.py pins to the reference's function
            from .train import contrastive_loss
            if not any(triples):
                return torch.tensor(0.0, requires_grad=True, device=features.device)
            return contrastive_loss(features, triples, self.temperature)
        total, count = 0.0, 0
        for b in range(B):
            for anchor, positive, negatives in triples[b]:
                a = feat[b, anchor].unsqueeze(0)
                logits = torch.cat([a @ feat[b, positive].unsqueeze(0).t(), a @ feat[b, negatives].t()], dim=-1) / self.temperature
                total = total + F.cross_entropy(logits, torch.tensor([0], dtype=torch.long, device=features.device))
                count += 1
        if count == 0:
            return torch.tensor(0.0, requires_grad=True, device=features.device)
        return total / count

    def compute_loss(self, original_image, generated_image, features, labels):
        """model/loss.py:111-126: content loss + contrastive loss."""
        return self.compute_content_loss(original_image, generated_image) + self.compute_contrastive_loss(features, labels)
