"""Input-pipeline contract of the hot path: cached-spectrogram dataset, language-ordered batch samplers,
collation into the tensors Tacotron.forward consumes, and the checkpoint dictionary.

What the reference does (behaviour restated, nothing imported from it):
  * meta-file `id|speaker|language|audio|mel.npy|lin.npy|text|phonemes`, one utterance per line (dataset/dataset.py:79-101);
    text cleaning flags and symbol ids `[_pad, _eos, _unk] + punctuations + alphabet`, EOS appended (utils/text.py:11-17,115-120);
    mel `.npy` files are `[num_mels, T]` and are normalised per channel when hp.normalize_spectrogram (utils/audio.py:105-108).
  * PerfectBatchSampler: position i of every mini-batch holds language `i mod G` so that the grouped encoders can reshape
    `[B, ...] -> [B/G, G*C, ...]`; batch size divisible by G * data-parallel devices (utils/samplers.py:50-122).
  * TextToSpeechCollate: zero-padded utterances / spectrograms, stop targets = 1 on the last hp.stop_frames frames of each
    sample, optional sort by text length (dataset/dataset.py:262-322).  The reference's sort branch crashes when
    hp.multi_language is set (`one_hot` undefined, 1-D `.size(1)`, :299-303); here the branch simply permutes the ids.
  * checkpoints: {'epoch','model','optimizer','scheduler','parameters','criterion'} (train.py:302-310).

MI355X-side differences: one process per GPU, so the sampler shards every GLOBAL mini-batch into `world` contiguous
per-rank chunks that each contain whole language groups in order (what DataParallel.scatter produced on one host).
Audio feature extraction, phonemisation and logging are outside the hot path (packages absent here): spectrograms must be
cached `.npy` files and texts must already be (phonemised) strings in the meta-file.
"""
import os
import random

import numpy as np
import torch

from .params.params import Params as hp

_PAD, _EOS, _UNK = '_', '~', '@'


def symbol_table(use_phonemes=None):
    """Symbol -> id in the reference's order (utils/text.py:16-17,117)."""
    use_phonemes = hp.use_phonemes if use_phonemes is None else use_phonemes
    symbols = [_PAD, _EOS, _UNK] + list(hp.punctuations_in) + list(hp.punctuations_out) + \
        list(hp.phonemes if use_phonemes else hp.characters)
    return {s: i for i, s in enumerate(symbols)}


def to_sequence(text, use_phonemes=None):
    """String -> ids, unknown symbols -> UNK, EOS appended (utils/text.py:115-120)."""
    table = symbol_table(use_phonemes)
    unk = table[_UNK]
    return [table.get(c, unk) for c in text] + [table[_EOS]]


def clean_text(text, is_phonemes):
    """hp.use_punctuation / hp.case_sensitive / hp.remove_multiple_wspaces handling of dataset/dataset.py:104-116."""
    if not hp.use_punctuation:
        drop = set(hp.punctuations_in) | set(hp.punctuations_out)
        text = ''.join(c for c in text if c not in drop)
    if not hp.case_sensitive and not is_phonemes:
        text = text.lower()
    if hp.remove_multiple_wspaces:
        text = ' '.join(text.split())
    return text


class MelDataset(torch.utils.data.Dataset):
    """Cached-spectrogram dataset over a reference meta-file; items are (speaker id, language id, token ids, mel, None)."""

    def __init__(self, meta_file, root_dir, known_unique_speakers=()):
        self.root_dir = root_dir
        self.unique_speakers = list(known_unique_speakers)
        seen = set(self.unique_speakers)
        self.items = []
        with open(meta_file, 'r', encoding='utf-8') as f:
            for line in f:
                tok = line.rstrip('\n').split('|')
                if len(tok) < 8 or tok[2] not in hp.languages:
                    continue
                if tok[1] not in seen:
                    seen.add(tok[1])
                    self.unique_speakers.append(tok[1])
                self.items.append({'id': tok[0], 'speaker': self.unique_speakers.index(tok[1]),
                                   'language': hp.languages.index(tok[2]), 'spectrogram': tok[4],
                                   'text': to_sequence(clean_text(tok[6], False), use_phonemes=False),
                                   'phonemes': to_sequence(clean_text(tok[7], True), use_phonemes=True)})

    def __len__(self):
        return len(self.items)

    def get_num_speakers(self):
        """dataset/dataset.py:160-163; train.py:239 sizes the speaker table and the adversarial classifier with it."""
        return len(self.unique_speakers)

    def frames(self, index):
        """Frame count of utterance `index` from the .npy header (no data is read); cached."""
        it = self.items[index]
        if 'frames' not in it:
            it['frames'] = int(np.load(os.path.join(self.root_dir, it['spectrogram']), mmap_mode='r').shape[1])
        return it['frames']

    def _raw(self, it):
        mel = np.load(os.path.join(self.root_dir, it['spectrogram']))
        assert mel.shape[0] == hp.num_mels, f'spectrogram has {mel.shape[0]} channels, expected {hp.num_mels}'
        return mel

    def get_normalization_constants(self):
        """Per-channel mean of means / mean of standard deviations over the collection, each [num_mels, 1]
        (dataset/dataset.py:165-176); train.py stores them in hp.mel_normalize_mean / _variance before the first batch."""
        mean, std = 0.0, 0.0
        for it in self.items:
            mel = self._raw(it)
            mean = mean + np.mean(mel, axis=1, keepdims=True)
            std = std + np.std(mel, axis=1, keepdims=True)
        return mean / len(self.items), std / len(self.items)

    def __getitem__(self, index):
        """`index` may be a pair (index, T_global) from the sharding samplers: T_global = longest spectrogram of the GLOBAL
        mini-batch, which every rank pads to (under DataParallel all replicas see the global max T)."""
        pad_to = None
        if isinstance(index, (tuple, list)):
            index, pad_to = index
        it = self.items[index]
        mel = self._raw(it)
        if hp.normalize_spectrogram:
            mel = (mel - hp.mel_normalize_mean) / hp.mel_normalize_variance
        item = (it['speaker'], it['language'], it['phonemes'] if hp.use_phonemes else it['text'], mel, None)
        return item if pad_to is None else item + (int(pad_to),)


def _shard(batch, rank, world, data_source=None):
    """Rank `rank`'s contiguous chunk of a GLOBAL mini-batch.  When the dataset can tell frame counts, every index travels
    with the global batch's longest spectrogram so that all ranks pad to the same T (the loss terms are means over B*M*T
    elements; the reference computes them on the gathered global batch, train.py:173-179 + modules/tacotron2.py:466-470)."""
    per = len(batch) // world
    mine = batch[rank * per:(rank + 1) * per]
    if data_source is not None and hasattr(data_source, 'frames') and world > 1:
        t_global = max(data_source.frames(i) for i in batch)
        return [(i, t_global) for i in mine]
    return mine


class PerfectBatchSampler(torch.utils.data.Sampler):
    """Language-ordered mini-batches: global position i holds language i mod G; every rank receives a contiguous chunk of
    whole groups.  `batch_size` is the GLOBAL batch; it must be divisible by G * world (utils/samplers.py:70-73)."""

    def __init__(self, data_source, languages, batch_size, data_parallel_devices=1, shuffle=True, drop_last=False,
                 rank=0, world=1, seed=0):
        G = len(languages)
        ways = G * data_parallel_devices * world
        assert batch_size % ways == 0, 'Batch size must be divisible by number of languages times the number of devices.'
        self._data = data_source
        self._by_language = [[] for _ in range(G)]
        for idx in range(len(data_source.items)):
            self._by_language[data_source.items[idx]['language']].append(idx)
        self._batch_size, self._G, self._ways = batch_size, G, ways
        self._shuffle, self._drop_last, self._rank, self._world = shuffle, drop_last, rank, world
        self._seed, self._epoch = seed, 0

    def set_epoch(self, epoch):
        self._epoch = epoch

    def _global_batches(self):
        order = [list(ix) for ix in self._by_language]
        if self._shuffle:      # identical permutation on every rank
            rng = random.Random(self._seed * 1000003 + self._epoch)
            for ix in order:
                rng.shuffle(ix)
        rounds = min(len(ix) for ix in order) if order else 0
        batch = []
        for r in range(rounds):
            batch += [ix[r] for ix in order]
            if len(batch) == self._batch_size:
                yield batch
                batch = []
        if not self._drop_last and batch:
            groups = len(batch) // self._G
            keep = (groups // (self._ways // self._G)) * (self._ways // self._G) * self._G
            if keep > 0:
                yield batch[:keep]

    def __iter__(self):
        for b in self._global_batches():
            yield _shard(b, self._rank, self._world, self._data)

    def __len__(self):
        per_lang = self._batch_size // self._G
        return min((len(ix) + per_lang - 1) // per_lang for ix in self._by_language)


class GlobalBatchSampler(torch.utils.data.Sampler):
    """Plain (not language-ordered) GLOBAL mini-batches, sharded per rank: the reference's
    `DataLoader(batch_size, shuffle / sampler=RandomImbalancedSampler)` branch (train.py:231-236) under one process per GPU.
    `balanced` draws indices with replacement, weight total / count(language) (utils/samplers.py:6-30); the draw is seeded by
    (seed, epoch) so that every rank sees the same global batches."""

    def __init__(self, data_source, batch_size, shuffle=True, balanced=False, drop_last=True, rank=0, world=1, seed=0):
        assert batch_size % world == 0, 'Batch size must be divisible by the number of devices.'
        self._data, self._n = data_source, len(data_source.items)
        self._batch_size, self._shuffle, self._balanced, self._drop_last = batch_size, shuffle, balanced, drop_last
        self._rank, self._world, self._seed, self._epoch = rank, world, seed, 0

    def set_epoch(self, epoch):
        self._epoch = epoch

    def _order(self):
        g = torch.Generator().manual_seed(self._seed * 1000003 + self._epoch)
        if self._balanced:
            return list(RandomImbalancedSampler(self._data, generator=g))
        if self._shuffle:
            return torch.randperm(self._n, generator=g).tolist()
        return list(range(self._n))

    def __iter__(self):
        order = self._order()
        for i in range(0, len(order), self._batch_size):
            b = order[i:i + self._batch_size]
            if len(b) < self._batch_size:
                if self._drop_last:
                    break
                b = b[:(len(b) // self._world) * self._world]      # equal shards on every rank
                if not b:
                    break
            yield _shard(b, self._rank, self._world, self._data)

    def __len__(self):
        if self._drop_last:
            return self._n // self._batch_size
        return (self._n + self._batch_size - 1) // self._batch_size


class RandomImbalancedSampler(torch.utils.data.Sampler):
    """With-replacement sampling with weight total / count(language) (utils/samplers.py:6-30)."""

    def __init__(self, data_source, generator=None):
        langs = [it['language'] for it in data_source.items]
        freq = {}
        for l in langs:
            freq[l] = freq.get(l, 0) + 1
        total = float(len(langs))
        self._sampler = torch.utils.data.WeightedRandomSampler([total / freq[l] for l in langs], len(langs), generator=generator)

    def __iter__(self):
        return iter(self._sampler)

    def __len__(self):
        return len(self._sampler)


class Collate:
    """Mini-batch tuple (utterances, utterance_lengths, mel, lin, mel_lengths, stop_targets, speakers, languages)."""

    def __init__(self, sort_by_text_length):
        self.sort_by_text_length = sort_by_text_length

    def __call__(self, batch):
        n = len(batch)
        pad_to = max((item[5] for item in batch if len(item) > 5), default=0)       # global max T (data-parallel shards)
        batch = [item[:5] for item in batch]
        text_len = torch.tensor([len(u) for _, _, u, _, _ in batch], dtype=torch.int64)
        mel_len = torch.tensor([m.shape[1] for _, _, _, m, _ in batch], dtype=torch.int64)
        speakers = torch.tensor([s for s, _, _, _, _ in batch], dtype=torch.int64) if hp.multi_speaker else None
        languages = torch.tensor([l for _, l, _, _, _ in batch], dtype=torch.int64) if hp.multi_language else None
        # ids index embedding tables on the device: validate them here, on the host, where it is free
        vocab = hp.symbols_count() + 3
        worst = max((max(u) for _, _, u, _, _ in batch if len(u)), default=0)
        if worst >= vocab:
            raise ValueError(f'symbol id {worst} outside the embedding table ({vocab} rows): use_phonemes / use_punctuation do not '
                             'match the meta-file')
        if speakers is not None and getattr(hp, 'speaker_number', 0) and int(speakers.max()) >= hp.speaker_number:
            raise ValueError(f'speaker id {int(speakers.max())} but hp.speaker_number = {hp.speaker_number}')
        order = list(range(n))
        if self.sort_by_text_length:
            text_len, idx = torch.sort(text_len, descending=True, stable=True)
            order = idx.tolist()
            mel_len = mel_len[idx]
            speakers = speakers[idx] if speakers is not None else None
            languages = languages[idx] if languages is not None else None
        T = max(int(mel_len.max()), pad_to)
        utterances = torch.zeros(n, int(text_len.max()), dtype=torch.int64)
        mels = torch.zeros(n, hp.num_mels, T, dtype=torch.float32)
        stops = torch.zeros(n, T, dtype=torch.float32)
        for row, i in enumerate(order):
            _, _, u, m, _ = batch[i]
            utterances[row, :len(u)] = torch.as_tensor(u, dtype=torch.int64)
            mels[row, :, :m.shape[1]] = torch.as_tensor(np.asarray(m), dtype=torch.float32)
            stops[row, max(m.shape[1] - hp.stop_frames, 0):] = 1
        return utterances, text_len, mels, None, mel_len, stops, speakers, languages


def batch_to_device(collated, device):
    """The 8-tuple of Collate -> the keyword batch used by bench.train_step (lin spectrograms are outside the hot path)."""
    u, ul, mel, _, ml, stop, spk, lang = collated
    to = lambda t: None if t is None else t.to(device, non_blocking=True)
    return dict(text=to(u), text_length=ul, target=to(mel), target_length=ml, stop=to(stop), speakers=to(spk), languages=to(lang))


def save_checkpoint(path, epoch, model, optimizer, scheduler, criterion):
    """Reference dictionary layout (train.py:302-310); `model` may be wrapped (the 'module.' prefix is dropped on load)."""
    torch.save({'epoch': epoch, 'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
                'scheduler': scheduler.state_dict() if scheduler is not None else {}, 'parameters': hp.state_dict(),
                'criterion': criterion.state_dict()}, path)


def load_checkpoint(path, model=None, optimizer=None, scheduler=None, criterion=None, map_location='cpu'):
    """Restore hyper-parameters first (they size the model), then whichever objects are given; returns the state dict."""
    from .utils import remove_dataparallel_prefix
    state = torch.load(path, map_location=map_location, weights_only=False)
    hp.load_state_dict(state['parameters'])
    if model is not None:
        model.load_state_dict(remove_dataparallel_prefix(state['model']))
    if optimizer is not None and state.get('optimizer'):
        optimizer.load_state_dict(state['optimizer'])
    if scheduler is not None and state.get('scheduler'):
        scheduler.load_state_dict(state['scheduler'])
    if criterion is not None and state.get('criterion') is not None:
        criterion.load_state_dict(state['criterion'])
    return state
