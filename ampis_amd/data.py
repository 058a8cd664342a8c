"""detectron2.data façade: DatasetCatalog / MetadataCatalog (notebook cells 13, 16, 18, 28; ampis/visualize.py:152) and the
loader pieces AmpisTrainer names (ampis/data_utils.py:24,171-175)."""
import types


class _DatasetCatalog(dict):
    def register(self, name, func):
        assert callable(func), "You must register a function with `DatasetCatalog.register`!"
        assert name not in self, f"Dataset '{name}' is already registered!"
        self[name] = func

    def get(self, name):
        try:
            f = self[name]
        except KeyError as e:
            raise KeyError(f"Dataset '{name}' is not registered! Available datasets are: {', '.join(self.keys())}") from e
        return f()

    def list(self):
        return list(self.keys())

    def remove(self, name):
        self.pop(name)

    @property
    def data(self):   # notebook cell 13 prints list(DatasetCatalog.data.keys())
        return self


class Metadata(types.SimpleNamespace):
    name = "N/A"

    def as_dict(self):
        return dict(self.__dict__)

    def set(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    def get(self, key, default=None):
        return getattr(self, key, default)


class _MetadataCatalog(dict):
    def get(self, name):
        assert len(name)
        if name not in self:
            self[name] = Metadata(name=name)
        return self[name]

    def list(self):
        return list(self.keys())

    def remove(self, name):
        self.pop(name)


DatasetCatalog = _DatasetCatalog()
MetadataCatalog = _MetadataCatalog()


class DatasetMapper:
    """Turns a dataset dict into model input (image read + resize).  Training-mode mapping (flip, polygon -> mask
    targets) belongs to the training path, which this round does not build: constructing with is_train=True works
    (ampis/data_utils.py:174 does so), calling it raises."""

    def __init__(self, cfg, is_train=True):
        self.cfg = cfg
        self.is_train = is_train

    def __call__(self, dataset_dict):
        if self.is_train:
            raise NotImplementedError("ampis_amd: the training-mode DatasetMapper is not built yet (SURVEY §8 f1)")
        from .engine.defaults import read_image_bgr
        d = dict(dataset_dict)
        d["image_bgr"] = read_image_bgr(d["file_name"])
        return d


def build_detection_test_loader(cfg, dataset_name, mapper=None):
    """Iterable with len(); yields one-image batches (list of one dict), like detectron2's test loader."""
    dicts = DatasetCatalog.get(dataset_name)
    mapper = mapper or DatasetMapper(cfg, False)

    class _Loader:
        def __len__(self):
            return len(dicts)

        def __iter__(self):
            for d in dicts:
                yield [mapper(d)]

    return _Loader()
