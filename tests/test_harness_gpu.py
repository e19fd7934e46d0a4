"""End-to-end run of the hot path through the counterpart harness (train.py:318-443) on a synthetic
SCARED-like sequence rendered by the HIP path: tracking must recover the camera motion, mapping must
raise PSNR, densification must keep the cloud consistent.  (The reference's own PSNR/ATE need the real
dataset and its CUDA rasteriser; neither exists here -- SURVEY.md s6.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_progressive_then_global_run_on_a_synthetic_sequence():
    from fsgs_amd import metrics
    from fsgs_amd.sequence import learner_from_first_frame, make_sequence
    from fsgs_amd.trainer import PoseTrack, Runner

    torch.manual_seed(0)
    W, H, n = 320, 256, 7  # >= 2 patches of 128 px (local_pearson_loss box is hard-coded, train.py:257)
    frames, cam = make_sequence(W, H, n, P=40000, seed=1)
    pc = learner_from_first_frame(frames, cam, ratio=0.25)
    poses = PoseTrack(n, "cuda")
    # row0_depth_quirk=False: the stored previous-frame depth is the full map.  (The reference stores ROW 0
    # broadcast over the image, train.py:343, which starves the flow loss; that faithful mode is exercised in
    # the second test below.)
    run = Runner(pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, row0_depth_quirk=False)
    run.progressive_run()
    # frame 4 is a TEST frame (i_test = idx[4::8]): an unchanged train.py never renders it, its stored depth stays zero
    # and the flow term of the next frame's tracking is the constant 0 (Runner.test_frame_quirks, train.py:333-343)
    assert list(frames.i_test) == [4] and float(frames.pred_depths[4].abs().max()) == 0.0
    flow_by_frame = {e[1]: e[4] for e in run.log if e[0] == "track"}
    assert flow_by_frame[5] == 0.0 and flow_by_frame[3] > 0.0 and flow_by_frame[6] > 0.0
    rpe_t, rpe_r, ate = run.eval_pose()
    # untracked baseline: every pose left at identity
    ident = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    gt = np.stack(frames.gt_w2c)
    step = np.mean([np.linalg.norm(gt[i + 1][:3, 3] - gt[i][:3, 3]) for i in range(n - 1)])
    assert np.isfinite([rpe_t, rpe_r, ate]).all()
    assert rpe_t < 0.25 * step, "tracking did not recover the motion: rpe_t %g vs GT step %g" % (rpe_t, step)
    assert ate < 0.25 * step
    psnr0 = run.validation()
    assert psnr0 > 35.0, psnr0
    assert run.last_validation["psnr"] == psnr0 and 0.9 < run.last_validation["ssim"] <= 1.0, run.last_validation
    P0 = pc.num_points
    run.iteration = 290  # next mapping iterations cross a densification boundary (iteration % 300 == 0)
    import tempfile

    with tempfile.TemporaryDirectory() as ckpt_dir:
        run.global_run(40, eval_every=20, model_path=ckpt_dir, save_every=16)
        # train.py:401-443: test-frame evaluation at iter % eval_every == 0, checkpoints at iter % save_every == save_every - 1
        assert [it for it, _ in run.eval_log] == [0, 20, 40] and all(m["psnr"] > 25 and m["ssim"] > 0.8 for _, m in run.eval_log)
        assert sorted(os.listdir(ckpt_dir)) == ["chkpnt15.pth", "chkpnt31.pth", "poses15.pth", "poses31.pth"]
        (model_args, it_saved) = torch.load(os.path.join(ckpt_dir, "chkpnt31.pth"), weights_only=False)
        assert it_saved == 31 and len(model_args) == 12  # GaussianModel.capture()'s tuple (scene/gaussian_model.py:86-98)
    assert pc.num_points != P0 and pc.num_points > 0
    for k, v in pc.params.items():
        assert torch.isfinite(v).all(), k
        assert pc.optimizer.state[v]["exp_avg"].shape == v.shape
    assert pc.variables["denom"].shape[0] == pc.num_points
    psnr1 = run.validation()
    assert psnr1 > 30.0
    print("synthetic sequence: rpe_t %.5f (GT step %.5f) rpe_r %.4f deg ate %.5f psnr %.2f -> %.2f P %d -> %d" % (
        rpe_t, step, rpe_r, ate, psnr0, psnr1, P0, pc.num_points))


def test_reference_faithful_mode_runs_with_the_row0_depth_quirk():
    from fsgs_amd.sequence import learner_from_first_frame, make_sequence
    from fsgs_amd.trainer import PoseTrack, Runner

    torch.manual_seed(0)
    W, H, n = 320, 256, 4
    frames, cam = make_sequence(W, H, n, P=40000, seed=2)
    pc = learner_from_first_frame(frames, cam, ratio=0.25)
    poses = PoseTrack(n, "cuda")
    run = Runner(pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=100)
    assert run.row0_depth_quirk
    run.progressive_run()
    d = frames.pred_depths[1]
    assert torch.equal(d[0], d[-1])  # row 0 broadcast to every row (train.py:343)
    m = run.eval_pose()
    assert np.isfinite(m).all()
    assert len(run.log) == n - 1 and all(np.isfinite(l[2]) for l in run.log)


def test_run_from_an_on_disk_sequence_then_resume_from_a_checkpoint(tmp_path):
    """The caller-side data formats: a sequence in the reference's directory layout feeds the
    harness, and a chkpnt/poses pair in the reference's tuple layout resumes it (train.py:107-111,371-376)."""
    from fsgs_amd import checkpoint, dataset
    from fsgs_amd.optim import FusedAdam
    from fsgs_amd.sequence import learner_from_first_frame, make_sequence, write_frames
    from fsgs_amd.trainer import PoseTrack, Runner

    torch.manual_seed(0)
    W, H, n = 320, 256, 5
    synth_frames, _ = make_sequence(W, H, n, P=40000, seed=3)
    root = str(tmp_path / "scared_demo")
    write_frames(root, synth_frames)
    frames = dataset.read_sequence(root, device="cuda")
    assert len(frames.colors) == n and frames.colors[0].shape == (3, H, W) and frames.colors[0].is_cuda
    assert (torch.stack(frames.colors) - torch.stack(synth_frames.colors).clamp(0, 1)).abs().max() <= 0.5 / 255 + 1e-6
    assert torch.allclose(torch.stack(frames.monodeps), torch.stack(synth_frames.monodeps), atol=1e-5)
    assert torch.equal(torch.stack(frames.flows_fw), torch.stack(synth_frames.flows_fw))
    np.testing.assert_allclose(frames.K, synth_frames.K, rtol=1e-6)
    cam = dataset.camera_from_frames(frames)

    pc = learner_from_first_frame(frames, cam, ratio=0.25)
    poses = PoseTrack(n, "cuda")
    run = Runner(pc, poses, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, row0_depth_quirk=False)
    run.progressive_run()
    rpe_t, rpe_r, ate = run.eval_pose()
    gt = np.stack(frames.gt_w2c)
    step = np.mean([np.linalg.norm(gt[i + 1][:3, 3] - gt[i][:3, 3]) for i in range(n - 1)])
    assert rpe_t < 0.25 * step and ate < 0.25 * step, (rpe_t, ate, step)

    checkpoint.save(str(tmp_path / "out"), run.iteration, pc, poses, frames.K)
    pc2 = learner_from_first_frame(frames, cam, ratio=0.05)  # a different cloud: restore must replace all of it
    poses2 = PoseTrack(n, "cuda")
    it, K = checkpoint.load(str(tmp_path / "out" / ("chkpnt%d.pth" % run.iteration)), pc2, poses2)
    assert it == run.iteration and isinstance(pc2.optimizer, FusedAdam) and pc2.num_points == pc.num_points
    np.testing.assert_array_equal(K, frames.K)
    for k in pc.params:
        assert torch.equal(pc.params[k], pc2.params[k]) and pc2.params[k].is_cuda
        s1, s2 = pc.optimizer.state[pc.params[k]], pc2.optimizer.state[pc2.params[k]]
        assert int(s1["step"]) == int(s2["step"]) and torch.equal(s1["exp_avg_sq"], s2["exp_avg_sq"])
    assert torch.equal(poses.r, poses2.r) and torch.equal(poses.t, poses2.t)
    # both copies take the same two 2-view mapping iterations: the restored Adam state continues the trajectory
    run2 = Runner(pc2, poses2, frames, tracking_iter=50, mapping_iter=30, first_mapping_iter=200, row0_depth_quirk=False)
    run2.iteration, run2.keyframes = run.iteration, list(run.keyframes)
    run.rng.seed(5)
    run2.rng.seed(5)
    for r in (run, run2):
        torch.manual_seed(9)
        r.mapping(n - 1, 2, progressive=True)
    for k in pc.params:
        a, b = pc.params[k], pc2.params[k]
        assert (a - b).abs().max() <= 1e-4 * a.abs().max() + 1e-7, k
