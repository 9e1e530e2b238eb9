"""CPU oracle for the Laplace-diffusion sampling path  --  TEST INFRASTRUCTURE ONLY.

This package is a plain-torch fp32 (CPU) restatement of the arithmetic the
reference (`/root/reference`, Lweihan/LDiffusion) delegates to
`diffusers==0.34.0` (environment.yml:42), plus the reference-owned glue around
it (sampler loops, Laplace noise, luma features, mask tail).

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` -- as the checker / the timed CPU baseline.  The product
(`ldiffusion_amd/`) never imports, links or executes anything from here; it
fails loudly when the HIP extension is missing instead of falling back.

PARITY PINNING STATUS
---------------------
* pinned against installed third-party code in this container (fixtures under
  tests/golden/, generator scripts in scripts/):
    - Laplace inverse-CDF sampling   vs torch.distributions.Laplace (R7)
    - PIL convert("L") luma          vs Pillow (R9)
    - numpy_to_pil uint8 rounding    vs numpy round-half-even (R8)
    - metrics / label LUTs / sampler call order
                                     vs the reference's own python, imported
                                        with stubbed third-party modules
    - ROI tile origins / Gaussian importance map (BASELINE config 4)
                                     vs the reference's vendored nnU-Net helpers
                                        (tests/golden/reference_tiling.json)
* PARITY UNPINNED for everything that lives in diffusers (UNet2DConditionModel,
  AutoencoderKL, PNDMScheduler, decode_latents): diffusers is not in
  /root/reference, not installed, there is no network and no SD-v1.5
  checkpoint, and the reference has no test or golden vector for this path
  (SURVEY.md section 4 / 8c).  Those functions restate the published
  diffusers-0.34.0 algorithm from its documented class structure; each one
  names the diffusers class it mirrors so an audit against real diffusers is
  mechanical.  Structure is checked by parameter count (859.5 M / 34.2 M /
  49.5 M for the SD-v1.5 configs).
"""
