"""Output formats of the generated dataset -- SURVEY.md §8(f) row 3.

Mirrors what the reference writes so that ``SIGNeRFDataParser`` (signerf/data/signerf_dataparser.py:99-170,210-228) reads it
unchanged:
  * ``tensor_to_image``  /root/reference/signerf/utils/image_tensor_converter.py:7-33 -- x*255 with a TRUNCATING uint8 cast
    (254.9/255 -> 254), done by a kernel; only the finished bytes cross PCIe;
  * directory layout    datasetgenerator.py:146-182 (images/ masks/ conditions/ rendered/ originals/ *_<f>/ references/);
  * transforms.json     datasetgenerator.py:286-295 (header) and :447-466 (one frame per saved view);
  * ``load_previous_experiment_cameras``  signerf/utils/load_previous_experiment_cameras.py:12-54 (the reader of the same file).
PNG encoding itself is PIL on the host, as in the reference.
"""

from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
from torch import Tensor

from . import _lib


def tensor_to_uint8(tensor: Tensor) -> Tensor:
    """[H,W,C] fp32 on the GPU -> [H,W,C] uint8 on the GPU, (uint8)(x * 255) truncating."""
    lib = _lib.load()
    t = tensor.detach().to(torch.float32).contiguous()
    with torch.cuda.device(t.device):
        out = torch.empty(t.shape, dtype=torch.uint8, device=t.device)
        _lib.check(lib.sn_tensor_to_uint8(_lib.ptr(t), t.numel(), _lib.ptr(out), _lib.current_stream()), None, "sn_tensor_to_uint8")
    return out


def tensor_to_image(tensor: Tensor):
    """image_tensor_converter.py:7-33: [H,W,3] -> RGB image, [H,W,1] -> 'L' image."""
    from PIL import Image

    assert len(tensor.shape) == 3, "Tensor must be of shape (H, W, C)"
    u8 = tensor_to_uint8(tensor).cpu().numpy()
    if tensor.shape[2] == 1:
        return Image.fromarray(u8.squeeze(), "L")
    assert tensor.shape[2] == 3, "Tensor must be of shape (H, W, 3)"
    return Image.fromarray(u8)


def image_to_tensor(image) -> Tensor:
    """image_tensor_converter.py:35-54."""
    import numpy as np

    if image.mode == "RGBA":
        image = image.convert("RGB")
    return torch.from_numpy(np.array(image, dtype="float32")) / 255.0


def load_previous_experiment_cameras(transforms_path: Union[str, Path]) -> Tuple[Tensor, Optional[Tensor], bool]:
    """load_previous_experiment_cameras.py:12-54: (reference c2w [R,3,4], synthetic c2w [S,3,4] | None, is_combined)."""
    with open(transforms_path) as f:
        transforms = json.load(f)
    frames = transforms["frames"]

    def stack(indices: List[int]) -> Tensor:
        return torch.stack([torch.tensor(frames[i]["scene_transform_matrix"][:3], dtype=torch.float32) for i in indices], dim=0)

    reference = stack(transforms["reference_indices"])
    synthetic = stack(transforms["generated_indices"]) if transforms.get("is_synthetic") else None
    return reference, synthetic, bool(transforms.get("is_combined", False))


def encode_png(u8, compress_level: int = 6) -> bytes:
    """A PNG file of an 8-bit [H,W,1] (greyscale) or [H,W,3] (RGB) array -- the same pixels `PIL.Image.save` writes, encoded WITHOUT the
    interpreter lock: numpy forms the filtered scanlines (filter "Up": each row minus the row above, row 0 unfiltered) and `zlib.compress`
    deflates them, both of which release the GIL, so a pool of host threads really encodes in parallel (Pillow's encoder holds the lock:
    measured r04, 16 / 32 / 64 / 96 threads all took 1.4 s for config 5's 470 files).  Lossless like any PNG; the BYTES differ from Pillow's
    (its filter heuristic and chunking), which no reader depends on (signerf/data/signerf_dataparser.py loads pixels)."""
    import struct
    import zlib

    import numpy as np

    h, w, c = u8.shape
    assert c in (1, 3) and u8.dtype == np.uint8 and h > 0 and w > 0
    flat = np.ascontiguousarray(u8).reshape(h, w * c)
    raw = np.empty((h, 1 + w * c), dtype=np.uint8)
    raw[0, 0] = 0
    raw[0, 1:] = flat[0]
    if h > 1:
        raw[1:, 0] = 2
        np.subtract(flat[1:], flat[:-1], out=raw[1:, 1:])   # modulo 256, as the format defines the filter

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0 if c == 1 else 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, int(compress_level))) + chunk(b"IEND", b""))


class GeneratedDataset:
    """Directory + transforms.json writer with the reference's layout and keys."""

    SUBDIRS = ("images", "masks", "conditions", "rendered", "originals")

    def __init__(self, path: Union[str, Path], dataset_name: str, downscale_factor: int = 2, write_images: bool = True, save_workers: int = 0,
                 png_compress_level: Optional[int] = None, png_encoder: str = "native"):
        """write_images=False: directories and transforms.json only (the bench's "PNG writes off" leg).  save_workers > 0: PNG encoding
        and the file writes run on that many host threads (zlib releases the GIL) while the caller goes on to the next view; the
        bytes of a file do not depend on it.  ``flush()`` waits for them.  png_compress_level: None = the reference's call
        (``Image.save(path)``, zlib level 6); 1 encodes ~2.3x faster into ~15 % larger files holding the same pixels.
        png_encoder: "native" = ``encode_png`` above (same pixels, encodes in parallel on the pool); "pil" = the reference's literal call."""
        if png_encoder not in ("native", "pil"):
            raise ValueError(f"png_encoder={png_encoder!r}: 'native' or 'pil' expected")
        self.png_encoder = png_encoder
        self.png_compress_level = png_compress_level
        self.dataset_path = Path(path) / dataset_name
        self.downscale_factor = downscale_factor
        self.write_images = write_images
        self.dirs: Dict[str, Path] = {}
        self._pool = None
        self._pending: List = []
        # at most this many encodes queued or running: the host copies of a view's 8 PNGs must not pile up when encoding is slower than
        # the loop; a failed write surfaces at the next save instead of at the end
        self._max_pending = 16 * max(1, save_workers)
        if save_workers > 0:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=save_workers)

    def close(self) -> None:
        """Waits for the pending writes and releases the worker threads (idempotent; also runs when the object is collected)."""
        pool, self._pool = self._pool, None
        try:
            self.flush()
        finally:
            if pool is not None:
                pool.shutdown(wait=True)

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover  (interpreter shutdown)
            pass

    def init_directory(self) -> None:
        """datasetgenerator.py:146-175 (config.yml is written by DatasetGenerator.init_directory)."""
        self.dataset_path.mkdir(parents=True, exist_ok=True)
        for name in self.SUBDIRS:
            for key, d in ((name, self.dataset_path / name), (f"{name}_scaled", self.dataset_path / f"{name}_{self.downscale_factor}")):
                d.mkdir(parents=True, exist_ok=True)
                self.dirs[key] = d
        self.dirs["references"] = self.dataset_path / "references"
        self.dirs["references"].mkdir(parents=True, exist_ok=True)
        self.transforms_path = self.dataset_path / "transforms.json"

    def save_image(self, tensor: Tensor, path: Path) -> None:
        """``tensor_to_image(tensor).save(path)``: the truncating uint8 cast on the GPU, one device-to-host copy of the bytes, PNG
        encoding on the host (in the pool when there is one)."""
        if not self.write_images:
            return
        from PIL import Image

        assert len(tensor.shape) == 3 and tensor.shape[2] in (1, 3), "Tensor must be of shape (H, W, 1) or (H, W, 3)"
        u8 = tensor_to_uint8(tensor).cpu().numpy()

        kw = {} if self.png_compress_level is None else {"compress_level": int(self.png_compress_level)}
        native = self.png_encoder == "native"
        level = 6 if self.png_compress_level is None else int(self.png_compress_level)

        def write():
            if native:
                data = encode_png(u8, level)
                with open(path, "wb") as f:
                    f.write(data)
            else:
                (Image.fromarray(u8.squeeze(), "L") if u8.shape[2] == 1 else Image.fromarray(u8)).save(path, **kw)

        if self._pool is None:
            write()
        else:
            while len(self._pending) >= self._max_pending:
                self._pending.pop(0).result()
            self._pending.append(self._pool.submit(write))

    def flush(self) -> None:
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()

    @staticmethod
    def new_transforms(original_transform_matrix: Tensor, original_scale_factor: float, is_synthetic: bool = False,
                       is_combined: bool = False) -> Dict[str, Any]:
        """datasetgenerator.py:286-295."""
        return {"camera_model": "OPENCV", "orientation_override": "none", "method": "SIGNeRF", "is_synthetic": is_synthetic,
                "is_combined": is_combined, "frames": [],
                "original_transform_matrix": original_transform_matrix.cpu().numpy().tolist(),
                "original_scale_factor": original_scale_factor}

    def save_generated_images(self, idx: int, images: Dict[str, Tensor], camera, current_transforms: Dict[str, Any],
                              is_original: bool = False) -> Dict[str, Any]:
        """datasetgenerator.py:398-468: PNGs by key + one frame appended to the transforms."""
        def save(key: str, directory: str, stem: str):
            if key in images:
                self.save_image(images[key], self.dirs[directory] / f"{stem}_{idx}.png")

        save("edited", "images", "image")
        save("render", "originals" if is_original else "rendered", "image")
        save("mask", "masks", "mask")
        save("condition", "conditions", "condition")
        save("edited_scaled", "images_scaled", "image")
        save("render_scaled", "rendered_scaled", "image")  # the reference writes both cases to rendered_<f> (:436-440)
        save("mask_scaled", "masks_scaled", "mask")
        save("condition_scaled", "conditions_scaled", "condition")
        c2w = torch.cat([camera.camera_to_worlds.cpu(), torch.tensor([[0.0, 0.0, 0.0, 1.0]])], dim=0).numpy().tolist()
        current_transforms["frames"].append({
            "fl_x": camera.fx.item(), "fl_y": camera.fy.item(), "cx": camera.cx.item(), "cy": camera.cy.item(),
            "w": camera.width.item(), "h": camera.height.item(),
            "file_path": f"./images/image_{idx}.png",
            "_mask_path": f"./masks/mask_{idx}.png",
            "transform_matrix": c2w,  # scene space, like the reference (FIXME at datasetgenerator.py:464)
            "scene_transform_matrix": c2w,
        })
        return current_transforms

    def write_transforms(self, transforms: Dict[str, Any]) -> None:
        self.flush()
        with open(self.transforms_path, "w", encoding="utf8") as f:
            json.dump(transforms, f, indent=4)
