"""Batch construction of the reference (host side): the step BEFORE the hot path.

Mirrors /root/reference/dataset_loaders/composite.py:18-126 (`MF`: windows of `steps` frames `skip` apart,
optionally with VOs appended; `MFOnline`: labelled window + unlabelled window with real VOs) and the pose
helpers they use from /root/reference/common/pose_utils.py: `calc_vos_simple` (:234-246), `calc_vos_safe`
(:219-232, :276-288 through the numpy `qexp`/`qlog` :306-327), `process_poses` (:329-356).  Everything here
is CPU data preparation in the reference too (DataLoader workers); the image datasets themselves
(7Scenes / RobotCar readers) are out of scope -- `MF` takes any indexable dataset of (image, pose6) pairs.
"""
import numpy as np
import torch

from .evaluate import qexp


def qlog(q):
    """logarithm map (4,) -> (3,)   (pose_utils.py:306-317)"""
    q = np.asarray(q)
    if all(q[1:] == 0):
        return np.zeros(3)
    return np.arccos(q[0]) * q[1:] / np.linalg.norm(q[1:])


def _qmult(a, b):
    """Hamilton product, normalised (pose_utils.py:44-62)"""
    w = a[0] * b[0] - np.dot(a[1:], b[1:])
    v = a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])
    q = np.hstack((w, v))
    return q / np.linalg.norm(q)


def _rotate(t, q):
    """t + 2 q_s (q_v x t) + 2 q_v x (q_v x t)   (pose_utils.py:120-132)"""
    b = np.cross(q[1:], t)
    return t + 2 * q[0] * b + 2 * np.cross(q[1:], b)


def calc_vos_simple(poses):
    """[N,T,6] -> [N,T-1,6]: differences of consecutive poses, all six components (pose_utils.py:234-246)"""
    poses = torch.as_tensor(poses)
    return poses[:, 1:] - poses[:, :-1]


def _calc_vos_safe_pairs(poses, pairs_of):
    poses = torch.as_tensor(poses)
    p = poses.detach().cpu().numpy().astype(np.float32)
    pairs = pairs_of(p.shape[1])
    out = np.zeros((p.shape[0], len(pairs), 6), dtype=np.float32)
    for n in range(p.shape[0]):
        for k, (i, j) in enumerate(pairs):
            q0 = np.asarray(qexp(p[n, i, 3:]), dtype=np.float32)
            q1 = np.asarray(qexp(p[n, j, 3:]), dtype=np.float32)
            q0i = np.hstack((q0[:1], -q0[1:]))
            out[n, k, :3] = _rotate(p[n, j, :3] - p[n, i, :3], q0i)
            out[n, k, 3:] = np.asarray(qlog(_qmult(q0i, q1).astype(np.float32)), dtype=np.float32)
    return torch.from_numpy(out).to(poses.dtype)


def calc_vos_safe(poses):
    """[N,T,6] -> [N,T-1,6]: VO of consecutive poses in the first pose's frame, through the numpy qexp/qlog
    (no gradient), fp32 as the reference (pose_utils.py:219-232, :276-288)"""
    return _calc_vos_safe_pairs(poses, lambda T: [(i, i + 1) for i in range(T - 1)])


def calc_vos_safe_fc(poses):
    """[N,T,6] -> [N,T(T-1)/2,6]: the same VO for ALL pairs i < j in lexicographic order (pose_utils.py:290-304);
    the `vo_func` of the fully connected pose graph (scripts/eval.py:120)"""
    return _calc_vos_safe_pairs(poses, lambda T: [(i, j) for i in range(T) for j in range(i + 1, T)])


def mat2quat(R):
    """rotation matrix -> unit quaternion (w, x, y, z), w >= 0 up to the caller's hemisphere fix (the
    reference calls transforms3d.quaternions.mat2quat, pose_utils.py:346; same eigen-free closed form)"""
    R = np.asarray(R, dtype=np.float64)
    K = np.array([[R[0, 0] - R[1, 1] - R[2, 2], 0, 0, 0],
                  [R[0, 1] + R[1, 0], R[1, 1] - R[0, 0] - R[2, 2], 0, 0],
                  [R[0, 2] + R[2, 0], R[1, 2] + R[2, 1], R[2, 2] - R[0, 0] - R[1, 1], 0],
                  [R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
    vals, vecs = np.linalg.eigh(K)  # transforms3d's method: eigenvector of the largest eigenvalue
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = -q
    return q


def process_poses(poses_in, mean_t, std_t, align_R, align_t, align_s):
    """raw 3x4 poses [N,12] -> aligned, normalised (translation, log-quaternion) [N,6]  (pose_utils.py:329-356)"""
    poses_in = np.asarray(poses_in)
    poses_out = np.zeros((len(poses_in), 6))
    poses_out[:, 0:3] = poses_in[:, [3, 7, 11]]
    for i in range(len(poses_out)):
        R = poses_in[i].reshape((3, 4))[:3, :3]
        q = mat2quat(np.dot(align_R, R))
        q *= np.sign(q[0])  # constrain to hemisphere
        poses_out[i, 3:] = qlog(q)
        t = poses_out[i, :3] - align_t
        poses_out[i, :3] = align_s * np.dot(align_R, t[:, np.newaxis]).squeeze()
    poses_out[:, :3] -= mean_t
    poses_out[:, :3] /= std_t
    return poses_out


def window_offsets(steps, skips, centre_shift=0):
    """Frame offsets of one window relative to its index: the running sum of the `steps - 1` gaps, re-centred so the
    middle frame (steps // 2; the reference's Python-2 integer division, composite.py:66) sits on the index, then shifted
    by `centre_shift` (no_duplicates: half a window, so that window 0 starts at frame 0)."""
    off = np.zeros(steps, dtype=np.int64)
    off[1:] = np.cumsum(np.asarray(skips, dtype=np.int64))
    return off - off[steps // 2] + centre_shift


class MF(torch.utils.data.Dataset):
    """Windows of `steps` frames `skip` apart around every index, optionally with VO targets: the batch layout contract
    of dataset_loaders/composite.py:18-102 (same constructor, `get_indices`, item shapes), pinned by golden vectors
    generated from the reference's own class (tests/golden/batch_construction.npz).  `dataset` is any indexable dataset
    returning (image [3,H,W], pose [6]); with include_vos and real, `gt_dataset` + `dataset.gt_idx` supply the absolute
    poses.  With a fixed skip the offsets are one precomputed table; with variable_skip every window draws its gaps from
    numpy's global generator exactly as the reference does (`np.random.randint(1, skip + 1, steps - 1)`)."""

    def __init__(self, dataset, include_vos=False, no_duplicates=False, steps=2, skip=1, variable_skip=False, real=False,
                 train=True, vo_func=calc_vos_simple, gt_dataset=None):
        if isinstance(dataset, str):
            raise NotImplementedError("the 7Scenes / RobotCar image readers are outside the MI355X hot path; pass a "
                                      "dataset object")
        self.steps, self.skip, self.variable_skip, self.real = steps, skip, variable_skip, real
        self.include_vos, self.train, self.vo_func, self.no_duplicates = include_vos, train, vo_func, no_duplicates
        self.dset, self.gt_dset = dataset, gt_dataset
        self.L = self.steps * self.skip
        self._shift = (steps // 2) * skip if no_duplicates else 0
        self._table = None if variable_skip else window_offsets(steps, [skip] * (steps - 1), self._shift)

    def get_indices(self, index):
        off = self._table
        if off is None:
            off = window_offsets(self.steps, np.random.randint(1, high=self.skip + 1, size=self.steps - 1), self._shift)
        return np.clip(index + off, 0, len(self.dset) - 1)  # windows at the sequence ends repeat the boundary frame

    @staticmethod
    def _frames(dataset, indices):
        images, poses = zip(*(dataset[int(i)] for i in indices))
        return torch.stack(images, dim=0), torch.stack(poses, dim=0)

    def __getitem__(self, index):
        idx = self.get_indices(index)
        imgs, poses = self._frames(self.dset, idx)
        if not self.include_vos:
            return imgs, poses
        vos = self.vo_func(poses.unsqueeze(0))[0]  # from THIS dataset's poses (possibly VO / noisy), :88
        if self.real:  # the absolute half of the target comes from the ground-truth dataset, :90-94
            _, poses = self._frames(self.gt_dset, [self.dset.gt_idx[i] for i in idx])
        return imgs, torch.cat((poses, vos), dim=0)

    def __len__(self):
        return len(self.dset) - ((self.steps - 1) * self.skip if self.no_duplicates else 0)


class MFOnline(torch.utils.data.Dataset):
    """MapNet++ batches (composite.py:104-126): a labelled window with absolute poses followed by an unlabelled window
    whose targets are real VOs -- images [2*steps,3,H,W], targets [steps + (steps-1), 6]; in gps_mode the second half
    carries absolute (GPS) poses instead: [2*steps, 6].  Length and indexing follow the unlabelled set; the labelled set
    is cycled."""

    def __init__(self, train_dataset, val_dataset, gps_mode=False, val_gt_dataset=None, **kwargs):
        self.gps_mode = gps_mode
        self.train_set = MF(train_dataset, train=True, **kwargs)
        self.val_set = MF(val_dataset, train=False, include_vos=(not gps_mode), real=True, vo_func=calc_vos_safe,
                          no_duplicates=True, gt_dataset=val_gt_dataset, **kwargs)

    def __getitem__(self, idx):
        lab_images, lab_poses = self.train_set[idx % len(self.train_set)]
        unl_images, unl_targets = self.val_set[idx % len(self.val_set)]
        # unl_targets = [absolute poses | VOs] (or absolute poses only in gps_mode): MapNet++ keeps the VO part
        tail = unl_targets if self.gps_mode else unl_targets[unl_images.shape[0]:]
        return torch.cat((lab_images, unl_images)), torch.cat((lab_poses, tail))

    def __len__(self):
        return len(self.val_set)


class SyntheticFrames(torch.utils.data.Dataset):
    """stand-in for the image datasets: L frames of N(0,1) pixels (the post-Normalize distribution) on a smooth
    random trajectory, poses as (translation, log-quaternion); `gt_idx` = identity, as a dataset whose `real`
    poses index the ground truth one-to-one"""

    # the statistics a `uint8=True` dataset is meant to be normalised with (model.set_input_u8(MEAN, STD))
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def __init__(self, length, H=256, W=341, seed=7, uint8=False):
        """`uint8=True`: frames as decoded images, uint8 [H,W,3] (what `model.set_input_u8` consumes: ToTensor and
        Normalize then run on the device and a frame crosses PCIe as 3 bytes per pixel instead of 12)"""
        g = torch.Generator().manual_seed(seed)
        self.images = None
        self.uint8 = uint8
        self.shape, self.seed, self.length = (3, H, W), seed, length
        t = torch.cumsum(0.05 * torch.randn(length, 3, generator=g), dim=0)
        axis = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0)
        ang = torch.cumsum(0.02 * torch.rand(length, generator=g), dim=0) + 0.05
        self.poses = torch.cat((t, axis[None, :] * (ang[:, None] / 2)), dim=1)
        self.gt_idx = np.arange(length)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(i))
        if self.uint8:
            _, H, W = self.shape
            return torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8), self.poses[int(i)]
        return torch.randn(*self.shape, generator=g), self.poses[int(i)]


class RenderedFrames(torch.utils.data.Dataset):
    """a LEARNABLE synthetic scene (SyntheticFrames' pixels say nothing about the pose): a frame is a smooth function of its
    camera pose -- K fixed gratings (orientation, spatial frequency) whose per-colour AMPLITUDES are smooth monotone functions of
    the 6-DoF pose, drawn with a random phase per frame, plus pixel noise --

        image[c, y, x] = sum_k A[k, c](pose) * sin(2 pi u[k] . (x / W, y / H) + phase[frame, k]) + noise * N(0, 1)
        A[k, c](pose)  = 0.6 + 0.4 tanh(M[k, c] . pose / scale)

    The pose is encoded in translation-invariant statistics (the energy per grating and colour), which a convolutional network
    with global average pooling can read, so a few hundred training steps regress it and the reference's evaluation metric
    (median translation / rotation error, scripts/eval.py:193-205) measures how well a training run did.  `scene_seed` fixes the
    scene (u, M) and the pose curve; `seed` the sample positions on the curve, the phases and the noise: a training set and a
    held-out set of the SAME scene use one scene_seed and two seeds.  Poses are (translation, log-quaternion) along a closed
    smooth curve, consecutive frames consecutive on it (windows of frames `skip` apart have small relative poses); `gt_idx` =
    identity.  All frames are rendered once, at construction."""

    def __init__(self, length, H=64, W=85, seed=7, scene_seed=1, K=6, noise=0.05):
        g = torch.Generator().manual_seed(seed)
        sg = torch.Generator().manual_seed(1000 + scene_seed)
        self.shape, self.seed, self.length = (3, H, W), seed, length
        # the pose curve (scene property) sampled at `length` consecutive positions with a per-set offset (seed)
        ph = 6.283185307 * torch.rand(5, generator=sg)
        s = (torch.arange(length, dtype=torch.float32) + torch.rand(1, generator=g)) / length * 6.283185307
        t = torch.stack((torch.sin(s + ph[0]), 0.8 * torch.sin(2 * s + ph[1]), 0.6 * torch.sin(3 * s + ph[2])), dim=1)
        axis0 = torch.nn.functional.normalize(torch.randn(3, generator=sg), dim=0)
        wob = torch.stack((torch.sin(s + ph[4]), torch.cos(2 * s + ph[4]), torch.sin(3 * s)), dim=1)
        axis = torch.nn.functional.normalize(axis0[None, :] + 0.8 * wob, dim=1)  # the rotation axis wanders around axis0
        ang = 0.8 + 0.5 * torch.sin(2 * s + ph[3])                               # 0.3 .. 1.3 rad
        self.poses = torch.cat((t, axis * (ang[:, None] / 2)), dim=1)
        self.gt_idx = np.arange(length)
        freq = (torch.rand(K, 2, generator=sg) * 4.0 + 2.0) * torch.sign(torch.randn(K, 2, generator=sg))  # cycles per image
        mix = torch.randn(K, 3, 6, generator=sg)
        scale = torch.tensor([1.0, 0.8, 0.6, 0.4, 0.4, 0.4])
        amp = 0.6 + 0.4 * torch.tanh(torch.einsum("kcj,nj->nkc", mix, self.poses / scale) * 0.7)  # [N, K, 3]
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32) / H, torch.arange(W, dtype=torch.float32) / W, indexing="ij")
        plane = 6.283185307 * (freq[:, 0, None, None] * xs[None] + freq[:, 1, None, None] * ys[None])  # [K, H, W]
        phase = 6.283185307 * torch.rand(length, K, generator=g)
        self.images = torch.empty(length, 3, H, W)
        for n0 in range(0, length, 64):  # (chunks: the [n, K, H, W] intermediate stays small)
            sl = slice(n0, min(length, n0 + 64))
            waves = torch.sin(plane[None] + phase[sl, :, None, None])
            self.images[sl] = torch.einsum("nkc,nkhw->nchw", amp[sl], waves) / K ** 0.5 * 1.5
        if noise > 0:
            self.images += noise * torch.randn(self.images.shape, generator=g)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        return self.images[int(i)], self.poses[int(i)]
