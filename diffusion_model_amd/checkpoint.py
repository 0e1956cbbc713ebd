"""Checkpoint contract of the reference: ``torch.save({'egnn':…, 'spectrum_compressor':…, 'gamma':…}, path)``
(main.py:219-226), loader ``load_model_state`` (parts/def_for_main.py:63-70, weights_only=True); the older
trainer writes the key 'GammaNetwork' instead of 'gamma' (train.py:358-366, test.py:149-157)."""
import torch


def save_model_state(nn_dict, path, params, diffusion_process=None):
    """main.py:219-226"""
    state = {"egnn": nn_dict["egnn"].state_dict()}
    if params.get("to_compress_spectrum"):
        state["spectrum_compressor"] = nn_dict["spectrum_compressor"].state_dict()
    if params.get("noise_schedule") == "learned":
        gamma = nn_dict.get("gamma", getattr(diffusion_process, "gamma", None))
        state["gamma"] = gamma.state_dict()
    torch.save(state, path)


def load_model_state(nn_dict, model_save_path, params):
    """load_model_state(nn_dict, model_save_path, params) of parts/def_for_main.py:63-70; never unpickles
    arbitrary objects (weights_only=True) and also accepts the older 'GammaNetwork' key."""
    state_dicts = torch.load(model_save_path, weights_only=True, map_location="cpu")
    nn_dict["egnn"].load_state_dict(state_dicts["egnn"])
    if params.get("to_compress_spectrum"):
        nn_dict["spectrum_compressor"].load_state_dict(state_dicts["spectrum_compressor"])
    if params.get("noise_schedule") == "learned":
        key = "gamma" if "gamma" in state_dicts else "GammaNetwork"
        nn_dict["gamma"].load_state_dict(state_dicts[key])
    return nn_dict
