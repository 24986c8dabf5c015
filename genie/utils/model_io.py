from genie2_amd.diffusion import (get_versions, get_epochs, load_config, load_default_model, load_model,  # noqa: F401
                                  load_pretrained_model, save_checkpoint)
