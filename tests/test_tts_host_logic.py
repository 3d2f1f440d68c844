"""CPU: TTSSession / TTSSndDispatch / actor facades with a recording worker (host logic only;
behaviour per Cluster/TTSSession.py:41-141, InfernTTSActor.py:21-52)."""
import torch

from infernos_amd.actors import InfernTTSActor, RemoteTTSSession
from infernos_amd.audio import AudioChunk
from infernos_amd.muxer import ASMarkerNewSent, ASMarkerSentDoneCB
from infernos_amd.tts import HelloSippyPlayRequest, TTSRequest, TTSSession, TTSSndDispatch, cleanup_text_eu


class FakeTTS:
    output_sr = 8000

    def __init__(self):
        self.reqs = []

    def infer(self, r):
        self.reqs.append(r)

    def get_voice(self, i):
        return torch.full((1, 512), float(i))

    def get_rand_voice(self):
        return torch.zeros(1, 512), 7

    def get_rand_voice_id(self):
        return 7


class FakeActor:
    def __init__(self):
        self.said = []
        self.tts_session_say = self

    def remote(self, rgen_id, req):
        self.said.append((rgen_id, req))
        return 'handle'


def test_say_dispatch_and_markers():
    tts, actr = FakeTTS(), FakeActor()
    sess = TTSSession(tts, actr)
    out = []
    sess.start(lambda chunk: out.append(chunk))
    rid = sess.say(TTSRequest('hello', speaker_id=3))
    assert rid in sess.active_req and len(tts.reqs) == 1
    pr = tts.reqs[0]
    assert isinstance(pr, HelloSippyPlayRequest) and pr.text == 'hello' and pr.session == sess.id
    assert float(pr.speaker[0, 0]) == 3.0
    pr.dispatch(torch.ones(100))
    pr.dispatch(torch.ones(50))
    pr.dispatch(None)
    assert [type(c) for c in out] == [AudioChunk, AudioChunk, ASMarkerNewSent]
    assert out[0].samplerate == 8000 and out[0].audio.numel() == 100
    assert rid not in sess.active_req                 # cleanup after the end-of-sentence marker
    # random voice when no speaker id is given
    req = TTSRequest('x')
    sess.say(req)
    assert req.speaker_id == 7


def test_multi_sentence_chains_through_actor():
    tts, actr = FakeTTS(), FakeActor()
    sess = TTSSession(tts, actr)
    out = []
    sess.start(lambda chunk: out.append(chunk))
    req = TTSRequest(['one', 'two', 'three'], speaker_id=1)
    sess.say(req)
    assert tts.reqs[0].text == 'one' and list(req.text) == ['two', 'three']
    tts.reqs[0].dispatch(None)
    assert isinstance(out[-1], ASMarkerSentDoneCB) and out[-1].sync
    assert out[-1].done_cb() == 'handle'              # self-RPC with the tail (TTSSession.py:113-115)
    assert actr.said == [(sess.id, req)]


def test_stop_saying_cancels():
    tts, actr = FakeTTS(), FakeActor()
    sess = TTSSession(tts, actr)
    out = []
    sess.start(lambda chunk: out.append(chunk))
    rid = sess.say(TTSRequest('abc', speaker_id=0))
    assert sess.stop_saying(rid) is True
    assert isinstance(out[-1], ASMarkerNewSent) and rid not in sess.active_req
    n = len(out)
    tts.reqs[0].dispatch(torch.ones(10))              # late audio of a cancelled request is dropped
    tts.reqs[0].dispatch(None)
    assert len(out) == n
    assert sess.stop_saying(rid) is False


def test_dispatch_rejects_empty_chunks():
    import pytest
    d = TTSSndDispatch(lambda chunk: None, 8000, None)
    with pytest.raises(AssertionError):
        d.sound_dispatch(torch.zeros(0))


def test_actor_facade_and_remote_session():
    a = InfernTTSActor()
    a.tts, a.tts_actr = FakeTTS(), FakeActor()
    r = RemoteTTSSession(a)
    got = []
    r.start(lambda chunk: got.append(chunk))
    rid = r.say(TTSRequest('hi', speaker_id=2))
    assert a.get_rand_voice_id() == 7 and r.sess_id in a.sessions
    assert r.stop_saying(rid) is True
    r.end()
    assert r.sess_id not in a.sessions


def test_cleanup_text():
    c = cleanup_text_eu()
    assert c('Ärger über Çà') == 'Erger yber Ca'
