"""Hot-path options with the reference's names and defaults
(options/seq2seqGAN_base_options.py:55-90, options/seq2seqGAN_train_options.py:35-58).
Any object with these attributes works (e.g. the reference's own argparse Namespace)."""
from types import SimpleNamespace

DEFAULTS = dict(
    hidden_size=256, word_vec_dim=300, n_layers=2, bidirectional=1, use_attention=1,
    decoder_max_len=5, encoder_max_len=17, operator_fc_dim=512, discrete_param=0, discrete_step=10,
    curve_steps=8, brightness_range=2, sharpness_range=1.5, exposure_range=3.5,
    saturation_range=(-0.2, 0.8), tone_curve_range=(0.5, 2), color_curve_range=(0.90, 1.10),
    input_dropout_p=0.2, dropout_p=0.2, variable_lengths=1, fix_input_embedding=1,
    start_id=1, end_id=2, null_id=0, explore_prob=0.05, learning_rate=1e-3, param_noise_factor=0.6,
    dataset='FiveK', session=1, vocab_dir='data/language', manual_seed=10,
    input_vocab_size=918, output_vocab_size=11, batch_size=64,
)


def default_options(**overrides):
    d = dict(DEFAULTS)
    d.update(overrides)
    return SimpleNamespace(**d)
