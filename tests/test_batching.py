"""CPU: the request-batching layer in front of the sampler on deterministic stand-in models --
`infer.infer_requests` (several infer_process calls as one batch), the multi-voice `[tag]` front-end (F/infer/infer_cli.py:181-208),
`serve.MicroBatcher` / `TTSManager(micro_batch=...)`, and `serve.ShardedSampler` + `rank_worker_loop` over gloo at world size 2."""
import os
import socket
import sys
import threading
import time
import wave

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_indic_server_f5_amd import infer, serve  # noqa: E402


class UnitModel:
    """Stand-in with the batch interface of F5HipModel: the mel of a unit is a closed form of ITS OWN prompt, tokens and frame count
    only, so any leak between units of a batch, or any re-ordering, changes the result."""
    device = torch.device("cpu")

    def __init__(self):
        self.batches = []
        self.mel_calls = 0

    def cond_mel(self, audio):
        self.mel_calls += 1
        n = audio.shape[-1] // 256 + 1
        return (audio[0, : (n - 1) * 256].reshape(n - 1, 256).mean(1, keepdim=True).repeat(1, 100)[None]
                if n > 1 else torch.zeros(1, 1, 100))

    def sample_units(self, audio, units, *, steps, cfg_strength, sway_sampling_coef, seed=None):
        audios = list(audio) if isinstance(audio, (list, tuple)) else [audio] * len(units)
        self.batches.append(len(units))
        out = []
        for a, (tokens, frames) in zip(audios, units):
            mel = self.cond_mel(a) if a.ndim == 2 else a
            key = float(mel.abs().sum()) * 1e-3 + sum(map(ord, "".join(tokens))) * 1e-4 + steps
            out.append(torch.linspace(0, 1, frames * 100).reshape(frames, 100) * key)
        return out


class Vocoder:
    def decode(self, mel):
        t = mel.shape[-1]
        return (torch.sin(torch.arange(256 * t, dtype=torch.float32) * 0.01) * mel.mean())[None]


def _clip(freq, seconds=3.0, amp=0.3):
    return (amp * torch.sin(2 * torch.pi * freq * torch.arange(int(24000 * seconds)) / 24000))[None], 24000


LONG = ("The quick brown fox jumps over the lazy dog. " * 6).strip()


def test_infer_requests_equals_one_infer_process_per_request():
    a, b = _clip(200.0), _clip(330.0, 4.0, 0.05)     # the second voice is below target_rms: gain up, restored after the vocoder
    reqs = [(a, "first voice words", LONG), (b, "second voice says", "Short one."), (a, "first voice words", "Again the first voice.")]
    m = UnitModel()
    got = infer.infer_requests(reqs, m, Vocoder(), nfe_step=4)
    assert len(m.batches) == 1 and m.batches[0] > len(reqs)          # ONE sampler call carrying every chunk of every request
    for (ra, rt, gt), (w, sr, spec) in zip(reqs, got):
        w1, sr1, spec1 = infer.infer_process(ra, rt, gt, UnitModel(), Vocoder(), nfe_step=4, show_info=lambda *_: None)
        assert sr == sr1 == 24000 and w.dtype == w1.dtype
        np.testing.assert_array_equal(w, w1)
        np.testing.assert_array_equal(spec, spec1)


def test_prepared_voice_is_computed_once():
    v = infer.PreparedVoice(_clip(250.0))
    m = UnitModel()
    infer.infer_requests([(v, "some words", "One."), (v, "some words", "Two.")], m, Vocoder(), nfe_step=4)
    infer.infer_requests([(v, "some words", "Three.")], m, Vocoder(), nfe_step=4)
    assert m.mel_calls == 1 and v.mel is not None                      # the reference latents of a voice are cached on the object


def test_split_voice_tags_and_multi_voice_concatenation():
    voices = {"main": dict(ref_audio=_clip(200.0), ref_text="main speaks"), "town": dict(ref_audio=_clip(300.0), ref_text="town speaks")}
    script = "A long time ago. [town] I live in town. [nobody] Who is this? [main]Back to me. [town]"
    pieces = infer.split_voice_tags(script, voices)
    assert pieces == [("main", "A long time ago."), ("town", "I live in town."), ("main", "Who is this?"), ("main", "Back to me.")]
    m = UnitModel()
    w, sr, specs = infer.infer_multi_voice(script, voices, m, Vocoder(), nfe_step=4)
    assert m.batches == [4] and sr == 24000 and len(specs) == 4
    parts = [infer.infer_process(voices[v]["ref_audio"], voices[v]["ref_text"], t, UnitModel(), Vocoder(), nfe_step=4, show_info=lambda *_: None)[0]
             for v, t in pieces]
    np.testing.assert_array_equal(w, np.concatenate(parts))             # plain concatenation between voices, like the reference's CLI
    with pytest.raises(ValueError):
        infer.infer_multi_voice("[town] hi", {"town": voices["town"]}, m, Vocoder())


def test_micro_batcher_batches_concurrent_requests_and_isolates_failures():
    seen = []

    def run(batch):
        seen.append(list(batch))
        time.sleep(0.05)                 # a batch "on the GPU": the next one forms meanwhile
        if any(r == "bad" for r in batch):
            raise ValueError("bad request")
        return [r.upper() for r in batch]

    mb = serve.MicroBatcher(run, max_requests=4, max_wait_ms=30)
    futs = [mb.submit(x) for x in ("a", "b", "c", "d", "e", "f")]
    assert [f.result(timeout=10) for f in futs] == ["A", "B", "C", "D", "E", "F"]
    assert max(mb.batch_sizes) == 4 and sum(mb.batch_sizes) == 6        # capped at max_requests, nothing lost
    f1, f2, f3 = mb.submit("x"), mb.submit("bad"), mb.submit("y")
    assert f1.result(timeout=10) == "X" and f3.result(timeout=10) == "Y"
    with pytest.raises(ValueError):
        f2.result(timeout=10)
    mb.close()
    with pytest.raises(RuntimeError):
        mb.submit("late")


def test_micro_batcher_close_under_load_resolves_every_future():
    """requests queued when close() is called are served (they were accepted), nothing lands behind the shutdown mark, a late
    submit is refused, and no future is left unresolved."""
    gate = threading.Event()

    def run(batch):
        gate.wait(timeout=10)
        return [r * 2 for r in batch]

    mb = serve.MicroBatcher(run, max_requests=2, max_wait_ms=1)
    futs = [mb.submit(i) for i in range(7)]
    closer = threading.Thread(target=mb.close)
    closer.start()
    time.sleep(0.05)
    with pytest.raises(RuntimeError):
        mb.submit(99)                    # refused as soon as close() has been entered, while batches are still running
    gate.set()
    closer.join(timeout=20)
    assert not closer.is_alive()
    assert [f.result(timeout=1) for f in futs] == [i * 2 for i in range(7)]
    # a worker thread that is gone: pending items are failed, not stranded
    mb2 = serve.MicroBatcher(lambda b: b, max_requests=2, max_wait_ms=1)
    mb2.close()
    mb2._q.put(("stranded", fut := serve.Future()))
    mb2._fail_pending()
    with pytest.raises(RuntimeError, match="closed"):
        fut.result(timeout=1)


def test_manager_without_batcher_serialises_the_device_path(tmp_path):
    """the routes run in a thread pool; with the default TTSManager() (no micro_batch) concurrent requests must still
    enter the model object one at a time -- the library allows one call in flight per handle."""
    class Probe(UnitModel):
        def __init__(self):
            super().__init__()
            self.inside, self.max_inside = 0, 0
            self.lock = threading.Lock()

        def sample_units(self, *a, **k):
            with self.lock:
                self.inside += 1
                self.max_inside = max(self.max_inside, self.inside)
            time.sleep(0.03)
            try:
                return super().sample_units(*a, **k)
            finally:
                with self.lock:
                    self.inside -= 1

    from fastapi.testclient import TestClient
    reg = serve.VoiceRegistry()
    reg.add("KAN_F (Happy)", _wav(tmp_path, "a.wav", 200), "reference words")
    model = Probe()
    mgr = serve.TTSManager(nfe_step=4).load(model, Vocoder())
    c = TestClient(serve.create_app(mgr, reg))
    out = [None] * 6

    def post(i):
        out[i] = c.post("/v1/audio/speech", json={"text": f"request {i} speaks."})

    th = [threading.Thread(target=post, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert all(r is not None and r.status_code == 200 for r in out)
    assert len(model.batches) == 6 and model.max_inside == 1
    mgr.close()
    assert mgr.model is None
    with pytest.raises(ValueError):
        mgr.synthesize("x", ref_audio_path="a", ref_text="b")


def _wav(tmp_path, name, freq):
    x = (6000 * np.sin(2 * np.pi * freq * np.arange(24000 * 3) / 24000)).astype(np.int16)
    p = tmp_path / name
    with wave.open(str(p), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(24000)
        f.writeframes(x.tobytes())
    return str(p)


def test_speech_route_batches_concurrent_requests(tmp_path):
    from fastapi.testclient import TestClient
    reg = serve.VoiceRegistry()
    reg.add("KAN_F (Happy)", _wav(tmp_path, "a.wav", 200), "reference words")
    reg.add("other", _wav(tmp_path, "b.wav", 320), "other reference")
    model = UnitModel()
    mgr = serve.TTSManager(nfe_step=4, micro_batch=dict(max_requests=8, max_wait_ms=200)).load(model, Vocoder())
    single = serve.TTSManager(nfe_step=4).load(UnitModel(), Vocoder())
    c = TestClient(serve.create_app(mgr, reg))
    texts = [f"request number {i} says hello." for i in range(6)]
    out = [None] * 6

    def post(i):
        name = "other" if i % 2 else "KAN_F (Happy)"
        out[i] = c.post("/v1/audio/speech/voice", json={"text": texts[i], "ref_audio_name": name})

    th = [threading.Thread(target=post, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert all(r is not None and r.status_code == 200 for r in out)
    assert max(mgr.batcher.batch_sizes) >= 2 and sum(mgr.batcher.batch_sizes) == 6     # requests met in the queue
    assert model.mel_calls == 2                                                         # one reference mel per voice, not per request
    for i, r in enumerate(out):                                                         # and each answer is the unbatched one
        v = reg.get("other" if i % 2 else "KAN_F (Happy)")
        ref = serve.wav_bytes(single.synthesize(texts[i], ref_audio_path=v.audio_path, ref_text=v.ref_text)).read()
        assert r.content == ref
    mgr.batcher.close()


# ---------------------------------------------------------------------------------------------------------------- world size 2
def _rank(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from tts_indic_server_f5_amd import infer as I, serve as S
    local = UnitModel()
    if rank != 0:
        n = S.rank_worker_loop(local)
        q.put((rank, n, list(local.batches)))
    else:
        sh = S.ShardedSampler(local)
        a, b = _clip(200.0), _clip(330.0, 4.0)
        reqs = [(a, "first voice words", LONG), (b, "second voice says", LONG + " And more."), (a, "first voice words", "Tail.")]
        got = I.infer_requests(reqs, sh, Vocoder(), nfe_step=4)
        ref = I.infer_requests(reqs, UnitModel(), Vocoder(), nfe_step=4)
        ok = all(np.array_equal(g[0], r[0]) and np.array_equal(g[2], r[2]) for g, r in zip(got, ref))
        got2 = I.infer_requests(reqs[:1], sh, Vocoder(), nfe_step=4)                   # a second job through the same loop
        ok = ok and np.array_equal(got2[0][0], ref[0][0])
        sh.close()
        q.put((rank, ok, list(local.batches)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sampler_gloo_world2_equals_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (x, b)) for r, x, b in (q.get(timeout=180) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
    assert res[0][0] is True                                  # rank 0: sharded == single-process, twice
    assert res[1][0] == 2                                     # rank 1 took part in both jobs, then was released
    assert sum(res[0][1]) > 0 and sum(res[1][1]) > 0          # both ranks sampled units


class FailingOnRank1(UnitModel):
    def sample_units(self, audio, units, **k):
        if dist.get_rank() == 1:
            raise RuntimeError("device fault on rank 1")
        return super().sample_units(audio, units, **k)


def _rank_fail(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from tts_indic_server_f5_amd import infer as I, serve as S
    local = FailingOnRank1()
    if rank != 0:
        n = S.rank_worker_loop(local)            # the failing job is reported through its gather; the loop survives it
        q.put((rank, n))
    else:
        sh = S.ShardedSampler(local)
        a = _clip(200.0)
        reqs = [(a, "first voice words", LONG), (a, "first voice words", LONG + " And more.")]
        mb = S.MicroBatcher(lambda rs: [w for w, _, _ in I.infer_requests(rs, sh, Vocoder(), nfe_step=4)], max_requests=4, max_wait_ms=200)
        futs = [mb.submit(r) for r in reqs]
        kinds = []
        for f in futs:
            try:
                f.result(timeout=60)
                kinds.append("ok")
            except S.ShardedJobError as e:
                kinds.append("sharded:" + str(e))
            except Exception as e:   # noqa: BLE001
                kinds.append("other:" + repr(e))
        refused = False
        try:
            sh.sample_units(a[0], [(["x"], 10)], steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0)
        except S.ShardedJobError:
            refused = True                        # no second job on a backend in an unknown state (and no new collective)
        mb.close()
        sh.close()
        q.put((rank, (kinds, list(mb.batch_sizes), refused)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_job_failure_on_one_rank_fails_the_batch_and_nobody_hangs():
    """a rank whose sampler raises still joins the job's gather; rank 0 raises ShardedJobError after the collective,
    MicroBatcher fails the whole batch WITHOUT per-request retries (a retry would issue a new broadcast against a backend in an unknown
    state), the sampler refuses later jobs, and the worker loop is released normally."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_fail, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kinds, sizes, refused = res[0]
    assert len(kinds) == 2 and all(k.startswith("sharded:") and "rank(s) [1]" in k for k in kinds), kinds
    assert sizes == [2] and refused                            # one batch, no per-request retry
    assert res[1] == 1                                         # the worker took part in exactly that job and was then released
