"""Named experiment presets = the override sets of the reference's `params/*.json`
and `params/singles/*.json`, expressed compositionally instead of shipped as json blobs.

`apply(name)` == `hp.load("params/<name>.json")` in the reference; `dump(dir)` writes the
json files for users who want the on-disk config surface back.  A CPU test
(tests/test_params.py) checks every preset against the reference's json when
/root/reference is present.
"""
import json
import os

from .params import Params, reset_defaults

_LATIN = " abcdefghijklmnopqrstuvwxyz"
_ACCENTS = "çèéßäöōǎǐíǒàáǔüèéìūòóùúāēěīâêôûñőű"
_CYRILLIC = "абвгдежзийклмнопрстуфхцчшщъыьэюяё"
_GREEK = "άέήίαβγδεζηθικλμνξοπρςíστυφχψωόύώ"
_WEST = "àâçèéêíôùûßäéöü"

ALPHABET_COMVOI = _LATIN + _ACCENTS + _CYRILLIC            # css_comvoi jsons (5 languages)
ALPHABET_CSS10 = ALPHABET_COMVOI + _GREEK                  # css10 jsons (10 languages)

CSS10_LANGUAGES = ["german", "french", "hungarian", "chinese", "spanish",
                   "dutch", "finnish", "russian", "japanese", "greek"]
COMVOI_LANGUAGES = ["de", "fr", "zh", "ru", "nl"]


def _multi(version, dataset, **kw):
    css10 = dataset == "css10"
    d = dict(balanced_sampling=True, batch_size=60 if css10 else 50, case_sensitive=False,
             characters=ALPHABET_CSS10 if css10 else ALPHABET_COMVOI, checkpoint_each_epochs=5,
             dataset=dataset, epochs=300,
             languages=list(CSS10_LANGUAGES if css10 else COMVOI_LANGUAGES),
             learning_rate=0.001, learning_rate_decay_each=10000, learning_rate_decay_start=10000,
             multi_language=True, predict_linear=False, version=version)
    if not css10:
        d.update(multi_speaker=True, speaker_embedding_dimension=32)
    d.update(kw)
    return d


def _single(version, language, characters, decay, batch_size=60):
    return dict(batch_size=batch_size, case_sensitive=False, characters=characters, dataset="css10",
                encoder_dimension=256, encoder_type="simple", epochs=300, languages=[language],
                learning_rate_decay_start=decay, learning_rate_decay_each=decay,
                multi_language=False, predict_linear=False, version=version)


_REVERSAL = dict(reversal_classifier=True, reversal_classifier_dim=256, reversal_gradient_clipping=0.25)
_SLOW_LR = dict(learning_rate=0.0001, learning_rate_decay_each=15000, learning_rate_decay_start=15000)

PRESETS = {
    # multilingual experiments (reference params/*.json)
    "generated_switching": _multi("GENERATED-SWITCHING", "css_comvoi", encoder_dimension=256,
                                  encoder_type="generated", generator_bottleneck_dim=4, generator_dim=10,
                                  language_embedding_dimension=0, perfect_sampling=True,
                                  reversal_classifier_w=0.125, **_REVERSAL),
    "generated_training": _multi("GENERATED-TRAINING", "css10", encoder_dimension=256,
                                 encoder_type="generated", generator_bottleneck_dim=8, generator_dim=20,
                                 language_embedding_dimension=32, perfect_sampling=True),
    "separate_switching": _multi("SEPARATE-SWITCHING", "css_comvoi", encoder_dimension=256,
                                 encoder_type="convolutional", language_embedding_dimension=0,
                                 perfect_sampling=True, reversal_classifier=False, **_SLOW_LR),
    "separate_training": _multi("SEPARATE-TRAINING", "css10", encoder_dimension=256,
                                encoder_type="convolutional", language_embedding_dimension=32,
                                perfect_sampling=True, reversal_classifier=False, **_SLOW_LR),
    "shared_switching": _multi("SHARED-SWITCHING", "css_comvoi", encoder_dimension=256,
                               encoder_type="simple", language_embedding_dimension=4,
                               reversal_classifier_w=0.5, **_REVERSAL),
    "shared_training": _multi("SHARED-TRAINING", "css10", encoder_type="simple",
                              language_embedding_dimension=32),
    # monolingual CSS10 models (reference params/singles/*.json; hu.json really says "greek")
    "singles/de": _single("DE", "german", _LATIN + _WEST, 2500),
    "singles/el": _single("EL", "greek", " " + _GREEK, 2500, batch_size=32),
    "singles/fi": _single("FI", "finnish", _LATIN + "äöü", 3000),
    "singles/fr": _single("FR", "french", _LATIN + _WEST, 5000),
    "singles/hu": _single("HU", "greek", _LATIN + "áéóúüöäőű", 2500),
    "singles/jp": _single("JP", "japanese", _LATIN, 3500),
    "singles/nl": _single("NL", "dutch", _LATIN + _WEST, 3500),
    "singles/ru": _single("RU", "russian", " " + _CYRILLIC, 3500),
    "singles/sp": _single("SP", "spanish", _LATIN + "áèéóúüöñí", 5000),
    "singles/zh": _single("ZH", "chinese", _LATIN + "ōǎǐíǒàáǔüèéìūòóùúāēěīâêôûñ", 2500),
}


def apply(name, reset=True, **extra):
    """Load preset `name` into the global Params (optionally after restoring defaults); `None` / "defaults" = the class defaults
    (the reference's LJ Speech configuration, which has no json)."""
    if reset:
        reset_defaults()
    if name not in (None, "defaults"):
        Params.load_state_dict(PRESETS[name])
    if Params.multi_language and not Params.language_number:
        Params.language_number = len(Params.languages)       # reference train.py:240
    Params.load_state_dict(extra)
    return Params


def dump(directory):
    """Write every preset as `<directory>/<name>.json` (the reference's on-disk layout)."""
    for name, overrides in PRESETS.items():
        path = os.path.join(directory, name + ".json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w", encoding="utf-8") as f:
            json.dump(overrides, f, indent=4, ensure_ascii=False, sort_keys=True)
