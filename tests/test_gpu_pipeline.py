"""GPU: the two-stream dataset loop (pipeline.restore_stream / ConditioningPrefetcher) returns, batch for batch, exactly
what encode -> restore back to back on one stream returns (no shared scratch between the streams, hand-over by event)."""
import pytest
import torch

from underwater_image_restoration_amd.synth import (daclip_vision_param_shapes, student_param_shapes, synth_state_dict,
                                                    synth_student_inputs)

pytestmark = pytest.mark.gpu
MK = dict(base_channels=32, num_levels=3, blocks_per_level=1, cond_dim=512, num_masks=4, num_degradation_types=5)


def test_restore_stream_equals_sequential():
    from underwater_image_restoration_amd.daclip import DaCLIPVision
    from underwater_image_restoration_amd.pipeline import ConditioningPrefetcher, restore_stream
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    model = NAFNetWFIGate(**MK, precision="bf16")
    model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=0))
    model = model.cuda().eval()
    clip = DaCLIPVision(precision="bf16")
    clip.load_state_dict(synth_state_dict(daclip_vision_param_shapes(), seed=1))
    clip = clip.cuda().eval()
    batches = []
    for i in range(4):
        x, _, m = synth_student_inputs(2, 128, 128, 512, 4, mask_hw=(7, 7), seed=40 + i)
        cin = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(60 + i))
        batches.append((x.cuda(), cin.cuda(), m.cuda()))
    want = []
    for x, cin, m in batches:
        _, z = clip.encode_image(cin, control=True)
        want.append(model(x, z, m)[0].clone())
    torch.cuda.synchronize()
    for _ in range(2):   # twice: the second pass reuses every buffer of the first
        got = [out[0].clone() for out in restore_stream(clip, model, batches)]
        torch.cuda.synchronize()
        assert len(got) == 4
        for g, w in zip(got, want):
            assert torch.equal(g, w)
    assert list(restore_stream(clip, model, [])) == []
    pf = ConditioningPrefetcher(clip)
    h = pf.launch(batches[0][1])
    with pytest.raises(RuntimeError):
        pf.launch(batches[1][1])                 # one encode in flight: the encoder's buffers are not double-buffered
    pf.take(h)


def test_student_branch_overlap_is_bit_identical():
    """SFFB on a second stream under WTB (student.overlap_branches) changes scheduling only: identical outputs, alphas and
    logits, run after run, in both precisions."""
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    for prec in ("bf16", "fp32"):
        model = NAFNetWFIGate(**MK, precision=prec)
        model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=3))
        model = model.cuda().eval()
        x, z, m = (t.cuda() for t in synth_student_inputs(3, 96, 160, 512, 4, mask_hw=(7, 7), seed=11))
        assert model.overlap_branches
        outs = []
        for flag in (True, False, True):
            model.overlap_branches = flag
            o, alphas, logits = model(x, z, m)
            torch.cuda.synchronize()
            outs.append((o.clone(), [a.clone() for a in alphas], logits.clone()))
        for o, alphas, logits in outs[1:]:
            assert torch.equal(o, outs[0][0]) and torch.equal(logits, outs[0][2])
            assert all(torch.equal(a, b) for a, b in zip(alphas, outs[0][1]))


def test_dual_replica_restorer_is_bit_identical():
    """Two forwards in flight on two buffer sets / streams: every batch's outputs equal the plain forward's, over two
    passes (buffers reused), with the consumer waiting on the hand-over event only."""
    from underwater_image_restoration_amd.pipeline import DualReplicaRestorer
    from underwater_image_restoration_amd.student import NAFNetWFIGate
    model = NAFNetWFIGate(**MK, precision="bf16")
    model.load_state_dict(synth_state_dict(student_param_shapes(**MK), seed=5))
    model = model.cuda().eval()
    batches = [tuple(t.cuda() for t in synth_student_inputs(2, 128, 96, 512, 4, mask_hw=(7, 7), seed=80 + i)) for i in range(5)]
    want = [tuple(model(*b)) for b in batches]
    want = [(o.clone(), [a.clone() for a in al], lg.clone()) for o, al, lg in want]
    torch.cuda.synchronize()
    dual = DualReplicaRestorer(model)
    assert dual.replicas[1] is not model and dual.replicas[1]._pack is model._pack      # shared packed weights
    assert all(p is q for p, q in zip(model.parameters(), dual.replicas[1].parameters()))
    for _ in range(2):
        handles = [dual.submit(*b) for b in batches]
        for h, (o, al, lg) in zip(handles, want):
            got_o, got_al, got_lg = dual.wait(h)
            assert torch.equal(got_o, o) and torch.equal(got_lg, lg)
            assert all(torch.equal(a, b) for a, b in zip(got_al, al))
    torch.cuda.synchronize()
