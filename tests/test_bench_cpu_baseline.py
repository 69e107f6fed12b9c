"""bench.cpu_baseline — the oracle (`"kind": "port"`) timed on the host cores beside the GPU number (task ④): a bounded sample,
extrapolated; for the Qwen2-Audio workloads it also times one audio-tower layer and the optimizer arithmetic so that the figure
estimates the whole step.  Here: the leg runs on a toy geometry, every term lowers the estimate, the sample says what it was."""
import types

import torch


def _toy(name="qwen2_audio_7b", wav=True):
    class AC:
        d_model, encoder_attention_heads, encoder_ffn_dim, encoder_layers = 64, 4, 128, 2

    class MC:
        audio_config = AC()

    class Cfg:
        hidden_size, intermediate_size, num_attention_heads, num_key_value_heads, head_dim = 64, 128, 4, 4, 16
        vocab_size, rms_norm_eps, rope_theta, rope_scaling, num_hidden_layers = 100, 1e-6, 1e4, None, 2

    return types.SimpleNamespace(seq_cfg=Cfg(), T=2048, B=2, name=name, model_config=MC(),
                                 wav=torch.zeros(3, 10) if wav else None)


def test_cpu_baseline_is_the_oracle_port_and_counts_tower_and_optimizer_for_the_audio_workloads():
    import bench
    threads = torch.get_num_threads()
    try:
        decoder_only = bench.cpu_baseline(_toy(), n_params=0)                  # (no parameter count: decoder + head only)
        whole = bench.cpu_baseline(_toy(), n_params=200_000_000)
        other = bench.cpu_baseline(_toy(name="llama_asr_1b", wav=False), n_params=200_000_000)
    finally:
        torch.set_num_threads(threads)
    for r in (decoder_only, whole, other):
        assert r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0 and "oracle fp32 eager" in r["sample"]
    assert "Whisper encoder layer" in whole["sample"] and "AdamW" in whole["sample"]
    assert "excluded -> an upper bound" in decoder_only["sample"] and "excluded -> an upper bound" in other["sample"]
    assert whole["value"] < 0.8 * decoder_only["value"]                       # the extra terms cost time
