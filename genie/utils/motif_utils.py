from genie2_amd.motif import load_motif_spec, sample_motif_mask, save_motif_pdb  # noqa: F401
