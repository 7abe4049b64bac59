"""speech_amd.loader -- the reference's data-loading API (/root/reference/speech/loader.py) for the drop-in drivers.

Host-side, I/O-bound code: nothing here is on the measured hot path, it exists so that
train.py / eval.py keep working against the same JSON-lines datasets and the same batch format
    batch = (inputs: tuple of float32 [T_i, F] arrays, labels: tuple of int lists)
Differences from the reference, all supersets:
  * the collate function returns a real tuple, not a one-shot zip iterator (loader.py:148 breaks eval_dev in py3);
  * wave files are read with scipy.io.wavfile when `soundfile` is not installed;
  * the character map is built from a SORTED set, so label ids do not depend on PYTHONHASHSEED.
"""
import json
import random

import numpy as np
import scipy.signal
import torch.utils.data as tud


def array_from_wave(file_name):
    """speech/utils/wave.py:8-10: (int16 samples, sample rate)."""
    try:
        import soundfile
        audio, samp_rate = soundfile.read(file_name, dtype="int16")
        return audio, samp_rate
    except ImportError:
        import scipy.io.wavfile
        samp_rate, audio = scipy.io.wavfile.read(file_name)
        return audio, samp_rate


def wav_duration(file_name):
    audio, samp_rate = array_from_wave(file_name)
    return audio.shape[0] / samp_rate


def log_specgram(audio, sample_rate, window_size=20, step_size=10, eps=1e-10):
    """loader.py:156-166: log power spectrogram, Hann window of window_size ms, hop step_size ms -> (frames, bins)."""
    nperseg = int(window_size * sample_rate / 1e3)
    noverlap = int(step_size * sample_rate / 1e3)
    _, _, spec = scipy.signal.spectrogram(audio, fs=sample_rate, window="hann", nperseg=nperseg, noverlap=noverlap,
                                          detrend=False)
    return np.log(spec.T.astype(np.float32) + eps)


def log_specgram_from_file(audio_file):
    audio, sr = array_from_wave(audio_file)
    return log_specgram(audio, sr)


def read_data_json(data_json):
    with open(data_json) as fid:
        return [json.loads(line) for line in fid]


def compute_mean_std(audio_files):
    feats = np.vstack([log_specgram_from_file(af) for af in audio_files])
    return np.mean(feats, axis=0), np.std(feats, axis=0)


class Preprocessor:
    END = "</s>"
    START = "<s>"

    def __init__(self, data_json, max_samples=100, start_and_end=True):
        """Feature statistics from up to max_samples files and the character <-> int maps (loader.py:15-45)."""
        data = read_data_json(data_json)
        audio_files = [d["audio"] for d in data]
        random.shuffle(audio_files)
        self.mean, self.std = compute_mean_std(audio_files[:max_samples])
        self._input_dim = self.mean.shape[0]
        chars = sorted(set(t for d in data for t in d["text"]))
        if start_and_end:
            chars.extend([self.END, self.START])  # START last: easy to exclude from a model's output classes
        self.start_and_end = start_and_end
        self.int_to_char = dict(enumerate(chars))
        self.char_to_int = {v: k for k, v in self.int_to_char.items()}

    def encode(self, text):
        text = list(text)
        if self.start_and_end:
            text = [self.START] + text + [self.END]
        return [self.char_to_int[t] for t in text]

    def decode(self, seq):
        text = [self.int_to_char[int(s)] for s in seq]
        if not self.start_and_end:
            return text
        s = 1 if text and text[0] == self.START else 0
        e = text.index(self.END) if text and text[-1] == self.END else len(text)
        return text[s:e]

    def preprocess(self, wave_file, text, device_features=False):
        """loader.py:65-69.  With `device_features` (make_loader(..., device_features=True) passes it through its
        dataset) the log spectrogram and the z-normalisation run on the GPU (speech_amd.features.log_specgram ->
        sa_log_specgram) and the features stay there as a float32 CUDA tensor: no host featuriser, no H2D copy of the
        features.  The switch belongs to the LOADER, never to this object: the Preprocessor is pickled next to the model
        (speech.save -> preproc.pyc) and eval.py re-uses it with forked DataLoader workers, which cannot touch the GPU."""
        if device_features:
            from . import features
            audio, sr = array_from_wave(wave_file)
            return features.log_specgram(audio, sr, mean=self.mean, std=self.std), self.encode(text)
        inputs = (log_specgram_from_file(wave_file) - self.mean) / self.std
        return inputs, self.encode(text)

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("device_features", None)  # round-2 objects carried the loader's switch: keep it out of checkpoints
        return state

    def __setstate__(self, state):
        state.pop("device_features", None)  # ... and ignore it in checkpoints written by round 2
        self.__dict__.update(state)

    @property
    def input_dim(self):
        return self._input_dim

    @property
    def vocab_size(self):
        return len(self.int_to_char)


class AudioDataset(tud.Dataset):
    """Examples bucketed by transcript length (bucket width 4), each bucket sorted by (duration, length)
    (loader.py:87-117), so that consecutive batches hold utterances of similar length."""

    def __init__(self, data_json, preproc, batch_size, device_features=False):
        data = read_data_json(data_json)
        self.preproc = preproc
        self.device_features = bool(device_features)
        bucket_diff = 4
        num_buckets = max(max(len(x["text"]) for x in data) // bucket_diff, 1)
        buckets = [[] for _ in range(num_buckets)]
        for d in data:
            buckets[min(len(d["text"]) // bucket_diff, num_buckets - 1)].append(d)
        for b in buckets:
            b.sort(key=lambda x: (round(x["duration"], 1), len(x["text"])))
        self.data = [d for b in buckets for d in b]

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        datum = self.data[idx]
        if self.device_features:
            return self.preproc.preprocess(datum["audio"], datum["text"], device_features=True)
        return self.preproc.preprocess(datum["audio"], datum["text"])


class BatchRandomSampler(tud.sampler.Sampler):
    """Consecutive batches, visited in random order without replacement (loader.py:120-137).

    A BATCH sampler (it yields one list of dataset indices per batch) with two additions for data-parallel training
    (SURVEY.md 8e):
      * its own random.Random: the seed is drawn ONCE, at construction, from Python's global `random` (which
        train.py seeds from the config, train.py:137), so every rank that builds its loaders in the same order visits
        the batches in the same order in every epoch -- whatever else consumes the global RNG in between (rank 0's
        dev-set pass, scheduled sampling);
      * rank r of `world` is handed utterances [r*B/W, (r+1)*B/W) of every global batch only, so a rank reads and
        featurises just its own shard (an empty list when the batch is smaller than the world).
    """

    def __init__(self, data_source, batch_size, world=1, rank=0, seed=None):
        it_end = len(data_source) - batch_size + 1
        self.batches = [list(range(i, i + batch_size)) for i in range(0, it_end, batch_size)]
        self.data_source = data_source
        self.world, self.rank = int(world), int(rank)
        self._rng = random.Random(random.getrandbits(64) if seed is None else seed)

    def __iter__(self):
        self._rng.shuffle(self.batches)
        for b in self.batches:
            base, rem = divmod(len(b), self.world)
            lo = self.rank * base + min(self.rank, rem)
            yield b[lo:lo + base + (1 if self.rank < rem else 0)]

    def __len__(self):
        return len(self.batches)


def _collate(batch):
    if not batch:
        return (), ()
    inputs, labels = zip(*batch)
    return inputs, labels


def make_loader(dataset_json, preproc, batch_size, num_workers=4, world=1, rank=0, device_features=False):
    """loader.py:139-150.  `world` / `rank`: this rank's shard of every global batch (see BatchRandomSampler).
    `device_features`: featurise on the GPU in the loading process itself (num_workers is forced to 0: forked
    DataLoader workers cannot use the device); batches then carry float32 CUDA tensors, which Model.collate pads on
    the device."""
    if device_features:
        num_workers = 0
    dataset = AudioDataset(dataset_json, preproc, batch_size, device_features=device_features)
    sampler = BatchRandomSampler(dataset, batch_size, world=world, rank=rank)
    return tud.DataLoader(dataset, batch_sampler=sampler, num_workers=num_workers, collate_fn=_collate)
