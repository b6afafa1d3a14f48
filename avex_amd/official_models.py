"""Packaged model descriptors (data): the reference's 10 official model ids with their
``model_spec`` values, default checkpoint URI and label-map URI
(reference data files: avex/api/configs/official_models/*.yml).

The ``beats``, ``efficientnet``, ``aves`` and ``eat_hf`` classes are built in this package (the EAT encoder's parity is
unpinned: its arithmetic is HF Hub remote code outside the reference tree, see oracle/eat_oracle.py).
"""
from __future__ import annotations

_RAW10 = dict(sample_rate=16000, representation="raw", normalize=False, target_length_seconds=10,
              window_selection="random")
_MEL = dict(sample_rate=16000, n_fft=800, hop_length=160, win_length=800, window="hann", n_mels=128,
            representation="mel_spectrogram", normalize=True, target_length_seconds=10, window_selection="random")


def _beats_init(**over):
    cfg = dict(
        activation_dropout=0.0, activation_fn="gelu", attention_dropout=0.0, conv_bias=False, conv_pos=128,
        conv_pos_groups=16, deep_norm=True, dropout=0.0, dropout_input=0.0, embed_dim=512,
        encoder_attention_heads=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072, encoder_layerdrop=0.05,
        encoder_layers=12, finetuned_model=True, gru_rel_pos=True, input_patch_size=16, layer_norm_first=False,
        layer_wise_gradient_decay_ratio=0.6, max_distance=800, num_buckets=320, predictor_class=527,
        predictor_dropout=0.0, relative_position_embedding=True, sample_frequency=16000.0, num_mel_bins=128,
        frame_length=25.0, frame_shift=10.0, fbank_mean=15.41663, fbank_std=6.55582)
    cfg.update(over)
    return cfg


def _hf(repo: str, file: str) -> str:
    return f"hf://EarthSpeciesProject/{repo}/{file}"


def _entry(repo, spec, labels=True):
    e = {"checkpoint_path": _hf(repo, repo + ".safetensors"), "model_spec": spec}
    if labels:
        e["class_mapping_path"] = _hf(repo, "label_map.json")
    return e


OFFICIAL_MODELS = {
    "esp_aves2_sl_beats_all": _entry("esp-aves2-sl-beats-all", dict(
        name="beats", pretrained=False, fine_tuned=True, device="cuda", init_config=_beats_init(), audio_config=_RAW10)),
    "esp_aves2_sl_beats_bio": _entry("esp-aves2-sl-beats-bio", dict(
        name="beats", pretrained=False, fine_tuned=True, device="cuda", init_config=_beats_init(), audio_config=_RAW10)),
    "esp_aves2_naturelm_audio_v1_beats": _entry("esp-aves2-naturelm-audio-v1-beats", dict(
        name="beats", pretrained=False, device="cuda", use_naturelm=True,
        init_config=_beats_init(attention_dropout=0.1, dropout=0.1, dropout_input=0.1, layer_wise_gradient_decay_ratio=1.0),
        audio_config=_RAW10), labels=False),
    "esp_aves2_effnetb0_all": _entry("esp-aves2-effnetb0-all", dict(
        name="efficientnet", pretrained=False, device="cuda", efficientnet_variant="b0", audio_config=_MEL)),
    "esp_aves2_effnetb0_bio": _entry("esp-aves2-effnetb0-bio", dict(
        name="efficientnet", pretrained=False, device="cuda", efficientnet_variant="b0", audio_config=_MEL)),
    "esp_aves2_effnetb0_audioset": _entry("esp-aves2-effnetb0-audioset", dict(
        name="efficientnet", pretrained=False, device="cuda", efficientnet_variant="b0", audio_config=_MEL), labels=False),
    "esp_aves2_eat_all": _entry("esp-aves2-eat-all", dict(
        name="eat_hf", pretrained=False, device="cuda", eat_norm_mean=-5.553, eat_norm_std=4.606, audio_config=_RAW10), labels=False),
    "esp_aves2_eat_bio": _entry("esp-aves2-eat-bio", dict(
        name="eat_hf", pretrained=False, device="cuda", eat_norm_mean=-5.553, eat_norm_std=4.606, audio_config=_RAW10), labels=False),
    "esp_aves2_sl_eat_all_ssl_all": _entry("esp-aves2-sl-eat-all-ssl-all", dict(
        name="eat_hf", pretrained=False, device="cuda", eat_norm_mean=-5.553, eat_norm_std=4.606, audio_config=_RAW10)),
    "esp_aves2_sl_eat_bio_ssl_all": _entry("esp-aves2-sl-eat-bio-ssl-all", dict(
        name="eat_hf", pretrained=False, device="cuda", eat_norm_mean=-5.553, eat_norm_std=4.606, audio_config=_RAW10)),
}
