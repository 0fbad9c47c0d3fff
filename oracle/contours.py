"""Oracle: the image stage of the merger (``segmentation/merger.py:219-333``) -- OpenCV calls restated.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: OpenCV (``cv2``) is a third-party dependency of the
reference that is absent from ``/root/reference`` and from this image (version unpinned: it comes with the nerfstudio
Docker image); what follows restates the PUBLISHED algorithm of ``cv::findContours`` (Suzuki & Abe 1985 as implemented in
``modules/imgproc/src/contours.cpp``: ``icvFindNextContour`` / ``icvFetchContour``), ``cv::contourArea`` (Green's formula),
``cv::boundingRect`` and ``cv::drawContours`` from recall, and is pinned only by the known-answer tests in
``tests/test_contours.py``.

What the reference does with them (per projected sub-cluster and camera):

``get_wo_occlusion_projection_area`` (``:249-271``): PNG -> gray -> ``threshold(thres, 255, THRESH_BINARY)`` (``> thres``) ->
``findContours(RETR_TREE, CHAIN_APPROX_SIMPLE)`` -> ``cnt = max(contours, key=contourArea)`` -> ``area = contourArea(cnt)``
(``< 10`` -> nothing) -> ``boundingRect(cnt)`` as (x, y, x + w, y + h).

``get_visible_projection_area`` (``:219-247``): the visible image cropped to that box -> same threshold / contours / max ->
``cv2.drawContours(segment_mask, cnt, contourIdx=-1, color=255, thickness=-1)``.  NOTE the second argument is ONE contour, not
a list of contours: OpenCV's binding reads an ``(N, 1, 2)`` array as N contours of one point each, so what is drawn
(filled) is the N VERTICES of the compressed border and nothing else.  ``area = segment_mask.sum()`` is therefore the number
of distinct vertex pixels, and the majority label is taken over those pixels: ``sorted([(count, label)], reverse=True)[0]``
(ties go to the larger label); ``label_area = 0`` for label 0.  Reproduced as written.
"""

from __future__ import annotations

from collections import Counter
from typing import List, Optional, Tuple

import numpy as np

EPS = 1e-6  # merger.py:22

# direction codes of icvCodeDeltas: 0 = east, then counter-clockwise on the screen (y grows downwards)
DX = (1, 1, 0, -1, -1, -1, 0, 1)
DY = (0, -1, -1, -1, 0, 1, 1, 1)


def threshold_binary(gray: np.ndarray, thres: int) -> np.ndarray:
    """``cv2.threshold(img, thres, 255, cv2.THRESH_BINARY)``: 255 where ``img > thres``."""
    return np.where(gray > thres, 255, 0).astype(np.uint8)


def _fetch_contour(img: np.ndarray, y0: int, x0: int, is_hole: bool, nbd: int) -> List[Tuple[int, int]]:
    """``icvFetchContour`` with CHAIN_APPROX_SIMPLE on the padded int image (0 background, 1 unvisited, other = marks).
    Returns the points (x, y) in padded coordinates."""
    pts: List[Tuple[int, int]] = []
    s_end = s = 0 if is_hole else 4
    found = False
    while True:
        s = (s - 1) & 7
        if img[y0 + DY[s], x0 + DX[s]] != 0:
            found = True
            break
        if s == s_end:
            break
    if not found:  # single pixel domain
        img[y0, x0] = -nbd
        return [(x0, y0)]
    y1, x1 = y0 + DY[s], x0 + DX[s]
    y3, x3 = y0, x0
    prev_s = s ^ 4
    while True:
        s_end = s
        while True:
            s += 1
            y4, x4 = y3 + DY[s & 7], x3 + DX[s & 7]
            if img[y4, x4] != 0:
                break
        s &= 7
        if ((s - 1) & 0xFFFFFFFF) < s_end:  # (unsigned)(s - 1) < (unsigned)s_end: the "right" bound was crossed
            img[y3, x3] = -nbd
        elif img[y3, x3] == 1:
            img[y3, x3] = nbd
        if s != prev_s:
            pts.append((x3, y3))
            prev_s = s
        if (y4, x4) == (y0, x0) and (y3, x3) == (y1, x1):
            break
        y3, x3 = y4, x4
        s = (s + 4) & 7
    return pts


def find_contours(binary: np.ndarray) -> List[Tuple[np.ndarray, bool]]:
    """``cv2.findContours(binary, RETR_TREE, CHAIN_APPROX_SIMPLE)[0]`` as a list of (points [N,2] int32 (x, y), is_hole) in
    the order OpenCV's Python binding returns the top of the tree: later-found siblings first (new contours are linked in
    front of their siblings).  (The nesting itself is not needed by the reference and is not built.)"""
    h, w = binary.shape
    img = np.zeros((h + 2, w + 2), dtype=np.int64)
    img[1:-1, 1:-1] = (binary != 0).astype(np.int64)
    found: List[Tuple[np.ndarray, bool]] = []
    nbd = 1
    for y in range(1, h + 1):
        prev = 0
        for x in range(1, w + 2):
            p = img[y, x]
            if p != prev:
                is_hole = False
                start = True
                if not (prev == 0 and p == 1):
                    if p != 0 or prev < 1:
                        start = False
                    else:
                        is_hole = True
                if start:
                    nbd += 1
                    pts = _fetch_contour(img, y, x - (1 if is_hole else 0), is_hole, nbd)
                    found.append((np.array(pts, dtype=np.int32) - 1, is_hole))  # back to image coordinates
                    p = img[y, x]
            prev = p
    return found[::-1]


def contour_area(pts: np.ndarray) -> float:
    """``cv2.contourArea(cnt)`` (oriented=False): |Green's formula| over the vertices."""
    if len(pts) == 0:
        return 0.0
    x = pts[:, 0].astype(np.float64)
    y = pts[:, 1].astype(np.float64)
    a = np.sum(np.roll(x, 1) * y - x * np.roll(y, 1))
    return float(abs(a) * 0.5)


def bounding_rect(pts: np.ndarray) -> Tuple[int, int, int, int]:
    x0, y0 = int(pts[:, 0].min()), int(pts[:, 1].min())
    return x0, y0, int(pts[:, 0].max()) - x0 + 1, int(pts[:, 1].max()) - y0 + 1


def largest_contour(binary: np.ndarray) -> Optional[np.ndarray]:
    cs = find_contours(binary)
    if not cs:
        return None
    return max((c for c, _ in cs), key=contour_area)  # Python's max: the first maximal element of the list


def wo_occlusion_projection_area(gray: np.ndarray, thres: int):
    """``get_wo_occlusion_projection_area`` (``:249-271``) on a gray image -> (area, bbox_xyxy | None)."""
    cnt = largest_contour(threshold_binary(gray, thres))
    if cnt is None:
        return EPS, None
    area = contour_area(cnt)
    if area < 10:
        return EPS, None
    x, y, w, h = bounding_rect(cnt)
    return area, (x, y, x + w, y + h)


def visible_projection_area(gray: np.ndarray, label_img: np.ndarray, bbox_xyxy, thres: int):
    """``get_visible_projection_area`` (``:219-247``) -> (area, label, label_area); see the module docstring for what
    ``drawContours(mask, cnt, -1, 255, -1)`` with a bare contour draws."""
    x0, y0, x1, y1 = bbox_xyxy
    crop = gray[y0:y1, x0:x1]
    cnt = largest_contour(threshold_binary(crop, thres))
    if cnt is None:
        return EPS, 0, EPS
    mask = np.zeros(crop.shape, dtype=bool)
    mask[cnt[:, 1], cnt[:, 0]] = True  # one filled single-point contour per vertex
    area = int(mask.sum())
    if area < 10:
        return EPS, 0, EPS
    labels = label_img[y0:y1, x0:x1][mask]
    ranked = sorted([(v, k) for k, v in Counter(labels.tolist()).items()], reverse=True)
    label_area, label = ranked[0]
    label_area = 0 if label == 0 else label_area
    return area, label, label_area


def process_super_cluster(wo_occ: np.ndarray, visible: np.ndarray, label_frames: np.ndarray, thres: int = 100,
                          frame_sampling_interval: int = 10, area_normalize: bool = False):
    """``process_super_cluster`` (``:273-333``) on arrays instead of a PNG tree: ``wo_occ`` / ``visible`` [n_cams, k, H, W]
    uint8 gray, ``label_frames`` [n_cams, H, W] uint8.  Returns the reference's ``cluster_prop`` dict."""
    n_cams, k = wo_occ.shape[:2]
    prop = {}
    for cid in range(k):
        vis_area = EPS * np.ones(n_cams)
        wo_area = EPS * np.ones(n_cams)
        overlap_area = EPS * np.ones(n_cams)
        overlap_label = np.zeros(n_cams)
        for cam in range(0, n_cams, frame_sampling_interval):
            area, bbox = wo_occlusion_projection_area(wo_occ[cam, cid], thres)
            wo_area[cam] = area
            if area == EPS:
                vis_area[cam] = area
                continue
            area, label, label_area = visible_projection_area(visible[cam, cid], label_frames[cam], bbox, thres)
            vis_area[cam], overlap_area[cam], overlap_label[cam] = area, label_area, label
        wo_norm = wo_area / wo_area.max()
        reliability = wo_norm * (overlap_area / wo_area) if area_normalize else np.ones_like(wo_area)
        prop[cid] = {"visible_area": vis_area, "wo_occ_area": wo_area, "wo_occ_area_norm": wo_norm, "label": overlap_label,
                     "label_overlap_area": overlap_area, "reliability": reliability}
    return prop
