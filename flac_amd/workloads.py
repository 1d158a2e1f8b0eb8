"""The benchmark configurations as data: what each one encodes or decodes, and the synthetic signal it runs on.

One place for `bench.py`, `tools/bench_configs.py` and the full-size GPU tests, so that a throughput figure and the
parity test beside it are about the same samples.  Every stream is `flac_amd.synth.synth_pcm` (three detuned tones plus
AR(2)-coloured noise per channel, channel 1 correlated with channel 0) with seed 1234 + 1000 * stream_id.
"""
from .synth import synth_pcm

BASE_SEED = 1234

CONFIGS = {
    # flac -5 on ten minutes of 16-bit stereo: the reference's own CPU-runnable case
    "C1": dict(level=5, channels=2, bps=16, rate=44100, seconds=600, over={}, settings={},
               title="C1: flac -5, 16-bit/44.1kHz stereo, blocksize 4096, 600 s"),
    # flac -8 on the same samples: the configuration the metric is quoted on
    "C2": dict(level=8, channels=2, bps=16, rate=44100, seconds=600, over={}, settings={},
               title="C2: flac -8 (max LPC order 12, tukey(0.5), Rice partition order 0-6), 16-bit/44.1kHz stereo, blocksize 4096, 600 s"),
    # flac -8 -e -p: exhaustive model search and a search over the coefficient precision, 137 candidates per subframe
    "C3": dict(level=8, channels=2, bps=24, rate=96000, seconds=600,
               over=dict(do_exhaustive_model_search=1, do_qlp_coeff_prec_search=1),
               settings={"set_do_exhaustive_model_search": 1, "set_do_qlp_coeff_prec_search": 1},
               title="C3: flac -8 -e -p, 24-bit/96kHz stereo, blocksize 4096, 600 s"),
    # eight independent channels (no mid/side beyond two channels)
    "C4": dict(level=8, channels=8, bps=24, rate=192000, seconds=60, over={}, settings={},
               title="C4: flac -8, 8 channels 24-bit/192kHz, blocksize 4096, 60 s"),
    # batch decode: 10 000 five-second -8 streams, 100 distinct signals x 100 copies
    "C5": dict(level=8, channels=2, bps=16, rate=44100, seconds=5, over={}, settings={}, streams=10000, distinct=100,
               title="C5: batch decode of 10 000 five-second flac -8 streams, 16-bit/44.1kHz stereo"),
}


def wide_samples(name, seconds=None):
    c = CONFIGS[name]
    return int(round((c["seconds"] if seconds is None else seconds) * c["rate"]))


_whole = {}   # the last whole stream made, by (rate, channels, bits, samples, seed)


def pcm(name, stream_id=0, seconds=None, wide=None, stream_wide=None):
    """Interleaved int32 PCM (wide, channels) of configuration `name`: the whole stream of signal `stream_id`, or its
    first `seconds` / `wide` samples.  The generator draws each channel's noise for the whole run at once, so a shorter
    run with the same seed is NOT a prefix of a longer one: the stream is always generated at its full length
    (`stream_wide`; the configuration's by default, or `wide` if that is longer) and cut, and the last one made is kept
    for the next call."""
    c = CONFIGS[name]
    n = wide if wide is not None else wide_samples(name, seconds)
    full = max(stream_wide if stream_wide is not None else wide_samples(name), n)
    key = (c["rate"], c["channels"], c["bps"], full, BASE_SEED + 1000 * stream_id)
    if key not in _whole:
        _whole.clear()
        _whole[key] = synth_pcm(wide_samples=full, sample_rate=c["rate"], channels=c["channels"], bits_per_sample=c["bps"], seed=key[4])
    whole = _whole[key]
    return whole if n == full else whole[:n]


def sample(name, seconds):
    """A short run of configuration `name`'s generator with the base seed: the same kind of signal as the stream (not a
    prefix of it, see pcm), made in proportion to its length.  For CPU baselines, which only need material of the kind."""
    c = CONFIGS[name]
    return synth_pcm(seconds=seconds, sample_rate=c["rate"], channels=c["channels"], bits_per_sample=c["bps"], seed=BASE_SEED)


def params(name, flac_amd_module):
    """The resolved encoder settings of configuration `name`."""
    c = CONFIGS[name]
    return flac_amd_module.preset(c["level"], c["channels"], c["bps"], c["rate"], **c["over"])


def container_bytes(name):
    return (CONFIGS[name]["bps"] + 7) // 8


def default_scaling(name):
    """The bench line's workload grows with the GPUs (its contract); the other configurations are one fixed job that is
    cut up: one stream's frames in contiguous ranges, or the batch's streams dealt round-robin."""
    return "weak" if name in ("C1", "C2") else "strong"


def plan(name, scaling, world, rank, blocksize, seconds=None, streams=None):
    """What rank `rank` of `world` does of configuration `name`.

    Encode configurations: the job is ONE stream, cut into contiguous frame ranges (flac_amd.shard.sample_range).
    "weak": the stream is world x the configuration's length, rank r's range is the r-th signal (stream_id r);
    "strong": the stream is the configuration's own, every rank takes its range of the same signal.
    C5: the streams are dealt round-robin (flac_amd.shard.deal_streams); "weak" deals world x the batch.
    """
    from . import shard
    c = CONFIGS[name]
    one = wide_samples(name, seconds)
    if name == "C5":
        total = (streams if streams else c["streams"]) * (world if scaling == "weak" else 1)
        return dict(kind="decode", total_streams=total, mine=shard.deal_streams(total, world, rank), distinct=min(c["distinct"], total),
                    stream_wide=one, job_samples=total * one * c["channels"])
    if scaling == "weak":
        frames_one = (one + blocksize - 1) // blocksize
        stream_wide = (world - 1) * frames_one * blocksize + one
    else:
        stream_wide = one
    first_frame, first_sample, wide = shard.sample_range(stream_wide, blocksize, world, rank)
    return dict(kind="encode", stream_wide=stream_wide, first_frame=first_frame, first_sample=first_sample, wide=wide,
                is_last=first_sample + wide == stream_wide, signal=rank if scaling == "weak" else 0,
                signal_offset=0 if scaling == "weak" else first_sample, signal_wide=wide if scaling == "weak" else stream_wide,
                job_samples=stream_wide * c["channels"])


def shard_pcm(name, pl):
    """The samples of an encode plan: int32 (wide, channels), cut out of the stream they belong to."""
    off = pl["signal_offset"]
    return pcm(name, stream_id=pl["signal"], wide=off + pl["wide"], stream_wide=pl["signal_wide"])[off:off + pl["wide"]]
