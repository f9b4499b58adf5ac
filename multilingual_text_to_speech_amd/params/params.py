"""Global hyper-parameter singleton (the `params.params.Params` config surface).

Mirrors the attribute names, defaults and static-method API of the reference's
`params/params.py:4-164` so that `from params.params import Params as hp`, json
overrides (`Params.load`), checkpoint round trips (`state_dict` / `load_state_dict`)
and `symbols_count()` behave identically.  Values are class attributes on purpose:
the reference mutates the class itself (`setattr(Params, k, v)`), and modules read
`hp.<name>` both at construction and at run time.
"""
import json


class Params:
    version = "1.0"

    # ---- training loop (reference params/params.py:12-34) -------------------------
    epochs = 300
    batch_size = 52
    learning_rate = 1e-3
    learning_rate_decay = 0.5
    learning_rate_decay_start = 15000
    learning_rate_decay_each = 15000
    learning_rate_encoder = 1e-3
    weight_decay = 1e-6
    encoder_optimizer = False
    max_output_length = 5000
    gradient_clipping = 0.25
    reversal_gradient_clipping = 0.25
    guided_attention_loss = True
    guided_attention_steps = 20000
    guided_attention_toleration = 0.25
    guided_attention_gain = 1.00025
    constant_teacher_forcing = True
    teacher_forcing = 1.0
    teacher_forcing_steps = 100000
    teacher_forcing_start_steps = 50000
    checkpoint_each_epochs = 10
    parallelization = True

    # ---- dataset (reference params/params.py:40-48) -------------------------------
    dataset = "ljspeech"
    cache_spectrograms = True
    languages = ['en-us']
    balanced_sampling = False
    perfect_sampling = False

    # ---- text (reference params/params.py:54-63) ----------------------------------
    characters = 'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz '
    case_sensitive = True
    remove_multiple_wspaces = True
    use_punctuation = True
    punctuations_out = '、。，"(),.:;¿?¡!\\'
    punctuations_in = '\'-'
    use_phonemes = False
    phonemes = 'ɹɐpbtdkɡfvθðszʃʒhmnŋlrwjeəɪɒuːɛiaʌʊɑɜɔx '

    # ---- model (reference params/params.py:69-119) --------------------------------
    embedding_dimension = 512
    encoder_type = "simple"            # simple | separate | shared | convolutional | generated
    encoder_dimension = 512
    encoder_blocks = 3
    encoder_kernel_size = 5
    generator_dim = 8
    generator_bottleneck_dim = 4
    prenet_dimension = 256
    prenet_layers = 2
    attention_type = "location_sensitive"
    attention_dimension = 128
    attention_kernel_size = 31
    attention_location_dimension = 32
    decoder_dimension = 1024
    decoder_regularization = 'dropout'  # dropout | zoneout
    zoneout_hidden = 0.1
    zoneout_cell = 0.1
    dropout_hidden = 0.1
    postnet_dimension = 512
    postnet_blocks = 5
    postnet_kernel_size = 5
    dropout = 0.5
    predict_linear = False
    cbhg_bank_kernels = 8
    cbhg_bank_dimension = 128
    cbhg_projection_kernel_size = 3
    cbhg_projection_dimension = 256
    cbhg_highway_dimension = 128
    cbhg_rnn_dim = 128
    cbhg_dropout = 0.0
    multi_speaker = False
    multi_language = False
    speaker_embedding_dimension = 32
    language_embedding_dimension = 4
    input_language_embedding = 4
    reversal_classifier = False
    reversal_classifier_type = "reversal"
    reversal_classifier_dim = 256
    reversal_classifier_w = 1.0
    stop_frames = 5
    speaker_number = 0                  # injected at run time (reference train.py:239)
    language_number = 0                 # injected at run time (reference train.py:240)

    # ---- audio (reference params/params.py:125-136) -------------------------------
    sample_rate = 22050
    num_fft = 1102
    num_mels = 80
    num_mfcc = 13
    stft_window_ms = 50
    stft_shift_ms = 12.5
    griffin_lim_iters = 60
    griffin_lim_power = 1.5
    normalize_spectrogram = True
    use_preemphasis = True
    preemphasis = 0.97

    # ---- io (reference params/params.py:139-164) ----------------------------------
    @staticmethod
    def load_state_dict(d):
        for key, value in d.items():
            setattr(Params, key, value)

    @staticmethod
    def state_dict():
        names = [a for a in dir(Params) if not a.startswith("__") and not callable(getattr(Params, a))]
        return {n: Params.__dict__[n] for n in names}

    @staticmethod
    def load(json_path):
        with open(json_path, 'r', encoding='utf-8') as f:
            Params.load_state_dict(json.load(f))

    @staticmethod
    def save(json_path):
        with open(json_path, 'w', encoding='utf-8') as f:
            json.dump(Params.state_dict(), f, indent=4)

    @staticmethod
    def symbols_count():
        n = len(Params.phonemes) if Params.use_phonemes else len(Params.characters)
        if Params.use_punctuation:
            n += len(Params.punctuations_out) + len(Params.punctuations_in)
        return n


_DEFAULTS = dict(Params.state_dict())


def reset_defaults():
    """Restore every attribute to its class default and drop run-time injected keys.

    Not in the reference (it has no tests and never needs to undo a json load); tests
    and bench.py switch between presets inside one interpreter and need it.
    """
    for name in list(Params.state_dict().keys()):
        if name not in _DEFAULTS:
            delattr(Params, name)
    Params.load_state_dict(_DEFAULTS)
