"""Image stage of the merger (``segmentation/merger.py:219-333``): known answers of the OpenCV restatement
(``oracle/contours.py``) on CPU, ``cn_contour_largest`` and the device ``process_super_cluster`` against it on the GPU."""

import numpy as np
import pytest
import torch

from oracle import contours as OC


def _img(h, w, *rects):
    b = np.zeros((h, w), np.uint8)
    for (y0, y1, x0, x1) in rects:
        b[y0:y1, x0:x1] = 255
    return b


# ------------------------------------------------------------------------------------------ known answers (OpenCV's outputs)
def test_find_contours_known_answers():
    # a filled 5 x 4 rectangle: its four corners, counter-clockwise on the screen, starting at the top-left pixel
    cs = OC.find_contours(_img(8, 10, (2, 6, 3, 8)))
    assert len(cs) == 1 and not cs[0][1]
    assert cs[0][0].tolist() == [[3, 2], [3, 5], [7, 5], [7, 2]]
    assert OC.contour_area(cs[0][0]) == 12.0  # (w - 1)(h - 1): the polygon runs through pixel centres
    assert OC.bounding_rect(cs[0][0]) == (3, 2, 5, 4)
    # a single pixel: one point, area 0
    cs = OC.find_contours(_img(3, 3, (1, 2, 1, 2)))
    assert [c.tolist() for c, _ in cs] == [[[1, 1]]] and OC.contour_area(cs[0][0]) == 0.0
    # a 3 x 3 ring: the hole border (found later, listed first) is the diamond of the four edge pixels
    ring = _img(5, 5, (1, 4, 1, 4))
    ring[2, 2] = 0
    cs = OC.find_contours(ring)
    assert [(c.tolist(), h) for c, h in cs] == [([[1, 2], [2, 1], [3, 2], [2, 3]], True),
                                                ([[1, 1], [1, 3], [3, 3], [3, 1]], False)]
    assert [OC.contour_area(c) for c, _ in cs] == [2.0, 4.0]  # the outer border always wins
    # an 8-connected diagonal: traced there and back, only the end points are direction changes
    d = np.zeros((5, 5), np.uint8)
    for i in range(3):
        d[i + 1, i + 1] = 255
    assert [c.tolist() for c, _ in OC.find_contours(d)] == [[[1, 1], [3, 3]]]
    # a plus sign: the border cuts the corners diagonally and never touches the centre
    p = np.zeros((5, 5), np.uint8)
    p[1, 2] = p[2, 1] = p[2, 2] = p[2, 3] = p[3, 2] = 255
    cs = OC.find_contours(p)
    assert cs[0][0].tolist() == [[2, 1], [1, 2], [2, 3], [3, 2]] and OC.contour_area(cs[0][0]) == 2.0
    # two blobs of equal area: OpenCV lists the later-found one first, and Python's max keeps the first maximum
    two = _img(12, 12, (1, 5, 1, 5), (6, 10, 6, 10))
    assert OC.bounding_rect(OC.largest_contour(two)) == (6, 6, 4, 4)
    # a blob touching the image border is traced like any other (the image is padded with background)
    assert OC.find_contours(_img(4, 4, (0, 2, 0, 3)))[0][0].tolist() == [[0, 0], [0, 1], [2, 1], [2, 0]]


def test_merger_image_functions_known_answers():
    gray = np.zeros((40, 60), np.uint8)
    gray[10:21, 20:41] = 200       # 21 x 11 block: area (20)(10) = 200
    gray[30:33, 5:8] = 255         # a 3 x 3 speck: area 4 < 10
    area, bbox = OC.wo_occlusion_projection_area(gray, 100)
    assert area == 200.0 and bbox == (20, 10, 41, 21)
    assert OC.wo_occlusion_projection_area(np.zeros((8, 8), np.uint8), 100) == (OC.EPS, None)
    speck = np.zeros((40, 60), np.uint8)
    speck[30:33, 5:8] = 255
    assert OC.wo_occlusion_projection_area(speck, 100) == (OC.EPS, None)
    assert OC.wo_occlusion_projection_area(np.full((8, 8), 100, np.uint8), 100) == (OC.EPS, None)  # > thres, not >=
    # visible projection: an octagon-like blob has many vertices; a rectangle has 4 (< 10 -> nothing)
    vis = np.zeros((40, 60), np.uint8)
    vis[10:21, 20:41] = 255
    labels = np.zeros((40, 60), np.uint8)
    assert OC.visible_projection_area(vis, labels, bbox, 100) == (OC.EPS, 0, OC.EPS)
    yy, xx = np.mgrid[0:40, 0:60]
    disc = ((yy - 15) ** 2 + (xx - 30) ** 2 <= 36).astype(np.uint8) * 255  # jagged border: 16+ direction changes
    labels[:, :30] = 3
    labels[:, 30:] = 7
    n, label, label_area = OC.visible_projection_area(disc, labels, bbox, 100)
    cnt = OC.largest_contour(OC.threshold_binary(disc[10:21, 20:41], 100))
    assert n == len({(int(x), int(y)) for x, y in cnt}) >= 10
    under = labels[10:21, 20:41][cnt[:, 1], cnt[:, 0]]
    counts = {int(v): int((under == v).sum()) for v in np.unique(under)}
    # the label under most vertices; on a tie the larger label (sorted reverse on (count, label))
    want = max(counts.items(), key=lambda kv: (kv[1], kv[0]))
    assert (label, label_area) == want
    labels[:] = 0  # background under every vertex: label 0, area 0
    assert OC.visible_projection_area(disc, labels, bbox, 100)[1:] == (0, 0)


# ------------------------------------------------------------------------------------------ device vs oracle
def _blob_image(rng, h, w, n_blobs, holes=True):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w), np.float32)
    for _ in range(n_blobs):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        ry, rx = rng.uniform(1.5, h / 4), rng.uniform(1.5, w / 4)
        img = np.maximum(img, np.clip(1.3 - ((yy - cy) / ry) ** 2 - ((xx - cx) / rx) ** 2, 0, 1))
    img += rng.uniform(0, 0.35, size=img.shape).astype(np.float32)  # speckle: ragged borders, pinholes, stray pixels
    if holes:
        for _ in range(2):
            cy, cx = int(rng.uniform(0, h)), int(rng.uniform(0, w))
            img[max(cy - 2, 0):cy + 2, max(cx - 2, 0):cx + 3] = 0
    return (np.clip(img, 0, 1) * 255).astype(np.uint8)


@pytest.mark.gpu
def test_hip_largest_contour_matches_the_opencv_restatement():
    from cropnerf_amd import ops

    rng = np.random.default_rng(0)
    H, W = 48, 64
    imgs = [_blob_image(rng, H, W, n) for n in (1, 2, 3, 5, 8) for _ in range(6)]
    imgs += [np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8), _img(H, W, (0, 3, 0, 4)), _img(H, W, (10, 11, 7, 8)),
             _img(H, W, (1, 5, 1, 5), (6, 10, 6, 10))]  # empty / full / on the border / one pixel / an area tie
    stack = np.stack(imgs)
    labels = rng.integers(0, 4, size=(3, H, W)).astype(np.uint8) * rng.integers(0, 2, size=(3, 1, 1)).astype(np.uint8)
    labels[1] = rng.integers(0, 6, size=(H, W))
    lidx = rng.integers(0, 3, size=len(imgs)).astype(np.int32)
    for thres in (100, 180):
        out = ops.contour_largest(torch.from_numpy(stack).cuda(), thres)
        area, bbox, start = out["area"].cpu().numpy(), out["bbox"].cpu().numpy(), out["start"].cpu().numpy()
        rois = []
        for j, g in enumerate(imgs):
            cnt = OC.largest_contour(OC.threshold_binary(g, thres))
            if cnt is None:
                assert area[j] == 0 and start[j] == -1, j
                rois.append((0, 0, 0, 0))
                continue
            assert area[j] == OC.contour_area(cnt), (thres, j)
            x, y, w, h = OC.bounding_rect(cnt)
            assert tuple(bbox[j]) == (x, y, w, h), (thres, j)
            assert start[j] == cnt[0][1] * W + cnt[0][0], (thres, j)  # the contour's first point = its first raster pixel
            rois.append((x, y, x + w, y + h))
        # second pass as the merger makes it: another image, restricted to the first one's box, labels under the vertices
        vis = np.roll(stack, 1, axis=0)
        roi = np.array(rois, np.int32)
        out2 = ops.contour_largest(torch.from_numpy(vis).cuda(), thres, roi=torch.from_numpy(roi).cuda(),
                                   labels=torch.from_numpy(labels).cuda(), label_index=torch.from_numpy(lidx).cuda())
        nv, lab, lc = (out2[k].cpu().numpy() for k in ("vertex_count", "label", "label_count"))
        for j in range(len(imgs)):
            x0, y0, x1, y1 = rois[j]
            crop = OC.threshold_binary(vis[j][y0:y1, x0:x1], thres)
            cnt = OC.largest_contour(crop) if crop.size else None
            if cnt is None:
                assert nv[j] == 0 and out2["area"][j].item() == 0, (thres, j)
                continue
            mask = np.zeros(crop.shape, bool)
            mask[cnt[:, 1], cnt[:, 0]] = True
            assert nv[j] == mask.sum(), (thres, j)
            under = labels[lidx[j]][y0:y1, x0:x1][mask]
            counts = {int(v): int((under == v).sum()) for v in np.unique(under)}
            assert (lab[j], lc[j]) == max(counts.items(), key=lambda kv: (kv[1], kv[0])), (thres, j)
            assert out2["area"][j].item() == OC.contour_area(cnt)


@pytest.mark.gpu
@pytest.mark.parametrize("area_normalize", [False, True])
def test_device_process_super_cluster_matches_the_restated_reference(area_normalize):
    """projections (float images, as a15 leaves them in HBM) -> cluster_prop -> affinity -> partition, device image stage
    against ``oracle/contours.process_super_cluster``; then through the (pinned) graph stage."""
    from cropnerf_amd.segmentation import merger

    rng = np.random.default_rng(5)
    n_cams, k, H, W = 23, 3, 60, 80
    yy, xx = np.mgrid[0:H, 0:W]
    wo = np.zeros((n_cams, k, H, W), np.float32)
    vis = np.zeros_like(wo)
    lab = np.zeros((n_cams, H, W), np.uint8)
    for c in range(n_cams):
        for s in range(k):
            if rng.uniform() < 0.2:
                continue  # this sub-cluster is out of this camera's view
            cy, cx, r = rng.uniform(15, H - 15), rng.uniform(15, W - 15), rng.uniform(5, 12)
            blob = np.clip(1.4 - ((yy - cy) ** 2 + (xx - cx) ** 2) / r ** 2, 0, 1) + rng.uniform(0, 0.2, size=(H, W))
            wo[c, s] = blob
            occl = xx > cx + rng.uniform(-r, r)  # something in front of part of it
            vis[c, s] = np.where(occl, 0, blob)
            lab[c][(yy - cy) ** 2 + (xx - cx) ** 2 <= (r * 1.5) ** 2] = rng.integers(0, 3) + (s // 2)  # covers the border pixels
    got = merger.process_super_cluster(torch.from_numpy(wo), torch.from_numpy(vis), lab, 100, 4, area_normalize)
    q = lambda a: merger.quantise_projection(torch.from_numpy(a)).numpy()
    ref = OC.process_super_cluster(q(wo), q(vis), lab, 100, 4, area_normalize)
    assert set(got) == set(ref) == set(range(k))
    for cid in range(k):
        for key in ("visible_area", "wo_occ_area", "wo_occ_area_norm", "label", "label_overlap_area", "reliability"):
            assert np.array_equal(np.asarray(got[cid][key], dtype=np.float64), np.asarray(ref[cid][key], dtype=np.float64)), (cid, key)
    assert any((ref[c]["wo_occ_area"] > 1).any() for c in range(k)) and any((ref[c]["label"] > 0).any() for c in range(k))
    aff = merger.calc_affinity(got)
    n, labels = merger.get_component(aff, "clique")
    assert aff.shape == (k, k) and 1 <= n <= k and len(labels) == k
    if not area_normalize:
        # the depth-projection merger's reliability (depth_projection_based_merger.py:263): label overlap / un-occluded area
        alt = merger.process_super_cluster(torch.from_numpy(wo), torch.from_numpy(vis), lab, 100, 4, False, plain_reliability="overlap")
        for cid in range(k):
            want = np.asarray(ref[cid]["label_overlap_area"], dtype=np.float64) / np.asarray(ref[cid]["wo_occ_area"], dtype=np.float64)
            assert np.array_equal(alt[cid]["reliability"], want), cid
            assert np.array_equal(got[cid]["reliability"], np.ones_like(want))


@pytest.mark.gpu
def test_merger_counts_from_a_projection_tree(tmp_path):
    """The reference's ``main`` flow on the PNG tree ``get_outputs_for_projections`` writes: two fruits, each split into two
    sub-clusters that carry the same instance label in every view -> each super-cluster merges into ONE fruit."""
    from PIL import Image

    from cropnerf_amd.segmentation import merger

    H, W, n_cams, k = 64, 96, 12, 2
    yy, xx = np.mgrid[0:H, 0:W]
    rng = np.random.default_rng(3)
    for sc in range(2):
        for cam in range(n_cams):
            d = tmp_path / "projection" / f"super_cluster_{sc}" / f"cam_{cam}"
            d.mkdir(parents=True)
            label = np.zeros((H, W), np.uint8)
            cx = 30 + 3 * cam
            for c in range(k):  # the two halves of one fruit, side by side
                blob = ((yy - 32) ** 2 + (xx - (cx + 14 * c)) ** 2 <= 81) & (rng.uniform(size=(H, W)) > 0.05)
                img = np.repeat((blob * 255).astype(np.uint8)[..., None], 3, -1)
                Image.fromarray(img).save(d / f"wo_occ_cluster_{c}.png")
                Image.fromarray(img).save(d / f"visible_cluster_{c}.png")
                label[(yy - 32) ** 2 + (xx - (cx + 14 * c)) ** 2 <= 144] = 5 + sc  # one instance covers both halves
            Image.fromarray(label).save(d / "label_frame_00001.png")
    total, counts, labels, affs = merger.count_from_projection_dir(tmp_path / "projection", 2, k, "clique",
                                                                   frame_sampling_interval=3)
    assert counts == [1, 1] and total == 2
    assert all(a[0, 1] > 0 for a in affs) and [sorted(set(l.tolist())) for l in labels] == [[1.0], [2.0]]
