"""Host-side Film: the reference's `Film` (pyrite/src/film.rs:9-114) as a numpy array of grains.

Layout is the reference's: grain index = (x + y*width)*bins + bin (film.rs:56), each grain {acc, weight} f32
(GrainData, film.rs:165-169), so `Film.grains` has shape [height, width, bins, 2]."""
import numpy as np

from . import abi


class Film:
    def __init__(self, width, height, grains_per_pixel=64, wavelength_span=(380.0, 780.0)):  # Film::new, film.rs:21-41
        self.width, self.height, self.bins = int(width), int(height), int(grains_per_pixel)
        self.wavelength_start = float(wavelength_span[0])
        self.wavelength_width = float(wavelength_span[1]) - float(wavelength_span[0])
        self.grains = np.zeros((self.height, self.width, self.bins, 2), dtype=np.float32)

    def desc(self):
        return abi.PyrFilmDesc(self.width, self.height, self.bins, self.wavelength_start, self.wavelength_width)

    def develop(self):
        """Grain::develop for every grain (film.rs:132-143): acc / weight where weight > 0, else 0. -> [h, w, bins]"""
        acc, weight = self.grains[..., 0], self.grains[..., 1]
        out = np.zeros_like(acc)
        np.divide(acc, weight, out=out, where=weight > 0)
        return out

    def total_weight(self):
        return float(self.grains[..., 1].sum(dtype=np.float64))
