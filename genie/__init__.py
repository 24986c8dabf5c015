"""Import-path compatibility with marvinli00/genie2: `genie.sampler.*`,
`genie.diffusion.*`, `genie.model.model.Denoiser`, `genie.utils.*` and the
`genie/sample_unconditional.py` CLI resolve to the MI355X-native
implementation in `genie2_amd`.  Only the denoising path is provided."""
