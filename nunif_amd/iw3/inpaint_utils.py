"""``FrameQueue`` of the video inpaint path.  Mirrors ``iw3/inpaint_utils.py`` ``FrameQueue`` :98-188: fixed-size device
buffers for the last ``seq`` warped eyes and their hole-logit maps; ``add`` / ``fill`` (repeat the last frame) / ``remove``
(slide by n) / ``get`` / ``clear``.  Pure buffer management — the tensors live in HBM, nothing is computed here."""
import torch


class FrameQueue:
    def __init__(self, synthetic_view, seq, height, width, dtype, device, mask_height=None, mask_width=None):
        mask_width = width if mask_width is None else mask_width
        mask_height = height if mask_height is None else mask_height
        self.left_eye = torch.zeros((seq, 3, height, width), dtype=dtype, device=device)
        self.right_eye = torch.zeros((seq, 3, height, width), dtype=dtype, device=device)
        new_mask = lambda: torch.zeros((seq, 1, mask_height, mask_width), dtype=dtype, device=device)   # noqa: E731
        self.left_mask = new_mask() if synthetic_view in ("both", "left") else None
        self.right_mask = new_mask() if synthetic_view in ("both", "right") else None
        self.synthetic_view = synthetic_view
        self.index = 0
        self.max_index = seq

    def full(self):
        return self.index == self.max_index

    def empty(self):
        return self.index == 0

    def add(self, left_eye, right_eye, left_mask=None, right_mask=None):
        self.left_eye[self.index] = left_eye
        self.right_eye[self.index] = right_eye
        if left_mask is not None:
            self.left_mask[self.index] = left_mask
        if right_mask is not None:
            self.right_mask[self.index] = right_mask
        self.index += 1

    def fill(self):
        """Repeat the newest frame until the queue is full; returns how many copies were added."""
        pad = 0
        if self.full():
            return pad
        i = self.index - 1
        frame = dict(left_eye=self.left_eye[i].clone(), right_eye=self.right_eye[i].clone())
        if self.left_mask is not None:
            frame["left_mask"] = self.left_mask[i].clone()
        if self.right_mask is not None:
            frame["right_mask"] = self.right_mask[i].clone()
        while not self.full():
            pad += 1
            self.add(**frame)
        return pad

    def remove(self, n):
        if 0 < n < self.max_index:
            # the reference moves entries n .. 2n-1 to the front (:167-175); with seq 12 and n = 6 that is the whole tail
            for buf in (self.left_eye, self.right_eye, self.left_mask, self.right_mask):
                if buf is not None:
                    buf[:n] = buf[n:2 * n].clone()
        self.index -= n
        assert self.index >= 0

    def get(self):
        if self.synthetic_view == "both":
            return self.left_eye, self.right_eye, self.left_mask, self.right_mask
        if self.synthetic_view == "left":
            return self.left_eye, self.right_eye, self.left_mask
        return self.left_eye, self.right_eye, self.right_mask

    def clear(self):
        self.index = 0
