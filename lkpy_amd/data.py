"""
Host-side data model: the subset of ``lenskit.data`` the hot-path components touch
(SURVEY.md section 2c).  Pure NumPy/SciPy plumbing; nothing here computes scores.

Mirrors (names, argument meaning, error behaviour):
``Vocabulary``  src/lenskit/data/_vocab.py:32-312
``ItemList``    src/lenskit/data/_items.py:46-1200
``RecQuery``    src/lenskit/data/_query.py
``Dataset``     src/lenskit/data/_dataset.py + ``MatrixRelationshipSet.scipy``
                (src/lenskit/data/_relationships.py:603-657)
``SparseRowArray`` src/lenskit/data/matrix.py:318-539 -> :mod:`lkpy_amd.matrix` (Arrow)
"""

from __future__ import annotations

from typing import Any, Iterable, Literal

import numpy as np
import pandas as pd
import scipy.sparse as sps


class Vocabulary:
    "Sorted, unique entity identifiers <-> contiguous numbers (``_vocab.py:32-312``)."

    def __init__(self, ids: Iterable | None = None, name: str | None = None, *, reorder=True):
        arr = np.asarray([] if ids is None else ids)
        if reorder:
            arr = np.unique(arr)  # ids are de-duplicated and sorted (_builder.py:345-346)
        self._ids = arr
        self._index = pd.Index(arr)
        self.name = name

    @property
    def index(self) -> pd.Index:
        return self._index

    @property
    def size(self) -> int:
        return len(self._ids)

    def __len__(self):
        return len(self._ids)

    def __eq__(self, other):
        return isinstance(other, Vocabulary) and np.array_equal(self._ids, other._ids)

    def __hash__(self):
        return id(self)

    def number(self, term, missing: Literal["error", "none"] | None = "error"):
        try:
            num = self._index.get_loc(term)
            return int(num)
        except KeyError:
            if missing == "error":
                raise
            return None

    def numbers(self, terms, missing: Literal["error", "negative"] = "error") -> np.ndarray:
        nums = np.require(self._index.get_indexer_for(np.asarray(terms)), dtype=np.int32)
        if missing == "error" and np.any(nums < 0):
            raise KeyError("unknown terms in vocabulary")
        return nums

    def id(self, num: int):
        return self._ids[num]

    def ids(self, nums=None) -> np.ndarray:
        return self._ids if nums is None else self._ids[np.asarray(nums)]

    def __repr__(self):
        return f"<Vocabulary {self.name or ''}: {len(self)} ids>"


class ItemList:
    """
    A list of items with optional scores and named fields (``_items.py:46-1200``):
    ``ItemList(other, scores=...)``, ``ItemList(item_ids)``,
    ``ItemList(item_nums=..., vocabulary=...)``, ``from_vocabulary``, ``numbers``, ``ids``,
    ``scores``, ``field``, ``remove``, ``top_n``.
    """

    def __init__(self, source=None, *, item_ids=None, item_nums=None, vocabulary=None,
                 ordered: bool = False, scores=None, **fields):  # fmt: skip
        self._ids = None
        self._nums = None
        self._vocab = vocabulary
        self._fields: dict[str, np.ndarray] = {}
        self.ordered = ordered
        if isinstance(source, ItemList):
            self._ids, self._nums = source._ids, source._nums
            self._vocab = vocabulary if vocabulary is not None else source._vocab
            self._fields = dict(source._fields)
            self.ordered = ordered or False
        elif isinstance(source, pd.DataFrame):
            self._ids = source["item_id"].to_numpy()
            for c in source.columns:
                if c != "item_id":
                    self._fields["score" if c == "score" else c] = source[c].to_numpy()
        elif source is not None:
            self._ids = np.asarray(source)
        if item_ids is not None:
            self._ids = np.asarray(item_ids)
        if item_nums is not None:
            self._nums = np.require(item_nums, dtype=np.int32)
            if vocabulary is None:
                raise ValueError("item numbers require a vocabulary")
        n = len(self)
        if scores is not None:
            fields["score"] = scores
        for name, val in fields.items():
            if val is None:
                continue
            arr = np.asarray(val)
            if arr.ndim == 0:
                arr = np.full(n, arr)
            if len(arr) != n:
                raise ValueError(f"field {name} has {len(arr)} values, list has {n}")
            if name == "score":
                arr = np.require(arr, dtype=np.float32)
            self._fields[name] = arr

    @classmethod
    def from_vocabulary(cls, vocab: Vocabulary) -> "ItemList":
        return cls(item_nums=np.arange(len(vocab), dtype=np.int32), vocabulary=vocab)

    def __len__(self):
        if self._ids is not None:
            return len(self._ids)
        if self._nums is not None:
            return len(self._nums)
        return 0

    def ids(self) -> np.ndarray:
        if self._ids is None:
            if self._nums is None:
                return np.zeros(0, dtype=np.int64)
            self._ids = self._vocab.ids(self._nums)
        return self._ids

    def numbers(self, vocabulary: Vocabulary | None = None,
                missing: Literal["error", "negative"] = "error") -> np.ndarray:  # fmt: skip
        if vocabulary is None or (self._vocab is not None and vocabulary is self._vocab):
            if self._nums is None:
                if self._vocab is None:
                    raise RuntimeError("item numbers not available without a vocabulary")
                self._nums = self._vocab.numbers(self.ids(), missing=missing)
            return self._nums
        if self._nums is not None and self._vocab is not None and vocabulary == self._vocab:
            return self._nums
        return vocabulary.numbers(self.ids(), missing=missing)

    def scores(self) -> np.ndarray | None:
        return self._fields.get("score")

    def field(self, name: str, format: str | None = None):
        if name == "score":
            return self.scores()
        return self._fields.get(name)

    @property
    def vocabulary(self):
        return self._vocab

    def _take(self, sel, ordered=False) -> "ItemList":
        out = ItemList(vocabulary=self._vocab, ordered=ordered)
        out._ids = None if self._ids is None else self._ids[sel]
        out._nums = None if self._nums is None else self._nums[sel]
        out._fields = {k: v[sel] for k, v in self._fields.items()}
        return out

    def remove(self, *, ids=None, numbers=None) -> "ItemList":
        "Drop the given items (``_items.py:1083-1128``)."
        if numbers is not None:
            mine = self.numbers()
            keep = ~np.isin(mine, np.asarray(numbers))
        else:
            keep = ~np.isin(self.ids(), np.asarray(ids))
        return self._take(keep)

    def top_n(self, n: int | None = None, *, scores: str | None = None) -> "ItemList":
        """
        The ``n`` highest-scored items, NaN scores dropped, result ordered
        (``_items.py:942-998`` -> ``_accel.data.argtopn`` / ``argsort_descending``).
        """
        from ._accel import data as _accel_data

        sc = self.scores() if scores is None else self.field(scores)
        if sc is None:
            raise RuntimeError("cannot rank items without scores")
        if n is None or n < 0:
            picked = _accel_data.argsort_descending(sc)
        else:
            picked = _accel_data.argtopn(sc, n)
        return self._take(np.asarray(picked), ordered=True)

    def to_df(self) -> pd.DataFrame:
        df = pd.DataFrame({"item_id": self.ids()})
        for k, v in self._fields.items():
            df[k] = v
        if self.ordered:
            df["rank"] = np.arange(1, len(self) + 1)
        return df

    def __repr__(self):
        return f"<ItemList of {len(self)} items, fields {sorted(self._fields)}>"


class ItemListCollection:
    """
    Item lists keyed by named tuples -- the shape ``batch.recommend`` / ``BatchResults.output``
    hand back in the reference (``src/lenskit/data/_collection/_base.py:48-592``,
    ``_list.py:27-200``; ``src/lenskit/batch/_runner.py:157-191``): ``lookup(key)`` /
    ``lookup(user_id=...)``, ``items()`` / ``lists()`` / ``keys()``, ``len``, iteration over
    ``(key, list)`` pairs, positional ``[i]``, ``key_fields`` / ``key_type``, ``to_df()``
    (key columns + the lists' columns), ``total_items()``, ``from_dict``.  A list-backed
    collection with a dict index, like the reference's ``ListILC``.
    """

    def __init__(self, key=("user_id",), *, index: bool = True):
        from collections import namedtuple

        if isinstance(key, type):
            self._key_class = key
        else:
            if isinstance(key, str):
                key = (key,)
            self._key_class = namedtuple("ListKey", list(key))
        self._lists: list[tuple[tuple, ItemList]] = []
        self._index: dict | None = {} if index else None

    @classmethod
    def from_dict(cls, data, key=None) -> "ItemListCollection":
        ilc = cls(key if key is not None else ("user_id",))
        for k, il in data.items():
            if isinstance(k, tuple):
                ilc.add(il, *k)
            else:
                ilc.add(il, k)
        return ilc

    @property
    def key_fields(self) -> tuple:
        return tuple(self._key_class._fields)

    @property
    def key_type(self):
        return self._key_class

    @classmethod
    def from_arrays(cls, keys, item_nums: np.ndarray, scores: np.ndarray, vocabulary: Vocabulary,
                    key=("user_id",)) -> "ItemListCollection":
        """
        A collection over the [B x n] result arrays of a batched recommend call (item numbers with
        -1 padding, scores): the ``ItemList`` of a key is only built when somebody asks for it
        (``lookup`` / iteration / ``[i]``), so handing back ten thousand lists costs nothing per
        list.  ``to_df`` and ``total_items`` work on the arrays directly.
        """
        ilc = cls(key)
        ilc._lists = _LazyLists(ilc._key_class, keys, item_nums, scores, vocabulary)
        ilc._index_stale = True  # the key -> position dict is built by the first ``lookup``
        return ilc

    def _ensure_index(self):
        if getattr(self, "_index_stale", False) and self._index is not None:
            self._index_stale = False
            ll = self._lists
            fresh = {ll.key(pos): pos for pos in range(len(ll.raw_keys))}
            fresh.update(self._index)  # (lists added since are later: the last of equal keys wins)
            self._index = fresh

    def add(self, list: ItemList, *fields, **kwfields):
        key = self._key_class(*fields, **kwfields)
        self._lists.append((key, list))
        if self._index is not None:
            # equal keys: every list is kept, ``lookup`` returns the LAST one -- the reference's
            # ``ListILC._add`` (src/lenskit/data/_collection/_list.py:190-193, 203-227)
            self._index[key] = len(self._lists) - 1

    def lookup(self, *args, **kwargs) -> ItemList | None:
        if len(args) == 1 and not kwargs and isinstance(args[0], tuple):
            key = self._key_class(*args[0])
        else:
            key = self._key_class(*args, **kwargs)
        if self._index is None:
            raise TypeError("cannot look up on a collection without an index")
        self._ensure_index()
        pos = self._index.get(key)
        return None if pos is None else self._lists[pos][1]

    def items(self):
        return iter(self._lists)

    def lists(self):
        return (il for _k, il in self._lists)

    def keys(self):
        if isinstance(self._lists, _LazyLists):
            return self._lists.all_keys()
        return (k for k, _il in self._lists)

    def total_items(self) -> int:
        if isinstance(self._lists, _LazyLists):
            return self._lists.total_items()
        return sum(len(il) for _k, il in self._lists)

    def __len__(self):
        return len(self._lists)

    def __iter__(self):
        return iter(self._lists)

    def __getitem__(self, pos: int):
        "positional, like the reference: ``(key, list)`` of the ``pos``-th entry"
        return self._lists[pos]

    def to_df(self) -> pd.DataFrame:
        if isinstance(self._lists, _LazyLists):
            return self._lists.to_df(self.key_fields)
        frames = []
        for key, il in self._lists:
            df = il.to_df()
            for f, v in zip(reversed(self.key_fields), reversed(key)):
                df.insert(0, f, v)
            frames.append(df)
        if not frames:
            return pd.DataFrame(columns=[*self.key_fields, "item_id"])
        return pd.concat(frames, ignore_index=True)

    def __repr__(self):
        return f"<ItemListCollection of {len(self)} lists, key {self.key_fields}>"


class _LazyLists:
    """
    The ``(key, ItemList)`` sequence of an array-backed collection (``from_arrays``): behaves like
    the list ``ItemListCollection`` keeps, builds an entry on access (and remembers it).
    """

    def __init__(self, key_class, keys, item_nums, scores, vocabulary):
        self.key_class = key_class
        self.raw_keys = keys if isinstance(keys, np.ndarray) else list(keys)
        self.nums, self.scores, self.vocab = item_nums, scores, vocabulary
        self.extra: list = []  # lists appended later (``add``)
        self._made: dict = {}

    def key(self, pos: int):
        k = self.raw_keys[pos]
        if isinstance(k, self.key_class):
            return k
        if isinstance(k, tuple):
            return self.key_class(*k)
        return self.key_class(k.item() if isinstance(k, np.generic) else k)

    def all_keys(self):
        for pos in range(len(self.raw_keys)):
            yield self.key(pos)
        for k, _il in self.extra:
            yield k

    def __len__(self):
        return len(self.raw_keys) + len(self.extra)

    def _make(self, pos: int):
        hit = self._made.get(pos)
        if hit is None:
            row = self.nums[pos]
            keep = row >= 0
            hit = (self.key(pos), ItemList(item_nums=row[keep], vocabulary=self.vocab,
                                           scores=self.scores[pos][keep], ordered=True))
            self._made[pos] = hit
        return hit

    def __getitem__(self, pos):
        if isinstance(pos, slice):
            return [self[i] for i in range(*pos.indices(len(self)))]
        if pos < 0:
            pos += len(self)
        if pos >= len(self.raw_keys):
            return self.extra[pos - len(self.raw_keys)]
        return self._make(pos)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def append(self, entry):
        self.extra.append(entry)

    def total_items(self) -> int:
        return int((self.nums >= 0).sum()) + sum(len(il) for _k, il in self.extra)

    def to_df(self, key_fields) -> pd.DataFrame:
        keep = self.nums >= 0
        counts = keep.sum(axis=1)
        cols = {}
        for j, f in enumerate(key_fields):
            col = self.raw_keys if (j == 0 and isinstance(self.raw_keys, np.ndarray)
                                    and self.raw_keys.ndim == 1) else \
                np.asarray([self.key(p)[j] for p in range(len(self.raw_keys))])
            cols[f] = np.repeat(col, counts)
        cols["item_id"] = self.vocab.ids(self.nums[keep])
        cols["score"] = self.scores[keep]
        ends = np.cumsum(counts)
        cols["rank"] = np.arange(1, int(ends[-1]) + 1 if len(ends) else 1) - \
            np.repeat(ends - counts, counts)
        df = pd.DataFrame(cols)
        if self.extra:
            frames = [df]
            for key, il in self.extra:
                d2 = il.to_df()
                for f, v in zip(reversed(key_fields), reversed(key)):
                    d2.insert(0, f, v)
                frames.append(d2)
            df = pd.concat(frames, ignore_index=True)
        return df


class RecQuery:
    "A recommendation query (``_query.py``): user id and/or history items."

    def __init__(self, user_id=None, user_items: ItemList | None = None, **kw):
        self.user_id = user_id
        self.history_items = user_items if user_items is not None else kw.get("history_items")
        self.session_items = kw.get("session_items")
        self.context_items = kw.get("context_items")

    @property
    def user_items(self):
        return self.history_items

    @property
    def query_items(self) -> ItemList | None:
        "context -> session -> history precedence."
        for x in (self.context_items, self.session_items, self.history_items):
            if x is not None:
                return x
        return None

    @property
    def all_items(self):
        return self.query_items

    @classmethod
    def create(cls, data) -> "RecQuery":
        if data is None:
            return cls()
        if isinstance(data, RecQuery):
            return data
        if isinstance(data, ItemList):
            return cls(user_items=data)
        if isinstance(data, (int, str, bytes, np.integer, np.str_)):
            return cls(user_id=data.item() if isinstance(data, np.generic) else data)
        raise TypeError(f"invalid query input (type {type(data)})")


from .matrix import SparseRowArray  # noqa: E402,F401  (Arrow extension array; matrix.py)


class _Matrix:
    "``MatrixRelationshipSet`` (rows = users, columns = items)."

    def __init__(self, ds: "Dataset"):
        self._ds = ds

    @property
    def row_vocabulary(self):
        return self._ds.users

    @property
    def col_vocabulary(self):
        return self._ds.items

    def scipy(self, attribute: str | None = None, *, layout: str = "csr", legacy: bool = False):
        "``_relationships.py:603-657``: values 1.0f32 when no attribute is requested."
        ds = self._ds
        if attribute is None or (attribute == "count" and "count" not in ds._attrs):
            values = np.ones(ds.interaction_count, dtype=np.float32)
        else:
            if attribute not in ds._attrs:
                raise KeyError(f"interactions have no attribute {attribute}")
            values = ds._attrs[attribute]
        shape = (ds.user_count, ds.item_count)
        if layout == "coo":
            return sps.coo_array((values, (ds._rows, ds._cols)), shape=shape)
        return sps.csr_array((values, ds._cols, ds._indptr), shape=shape)

    def row_items(self, user_id) -> ItemList | None:
        ds = self._ds
        u = ds.users.number(user_id, missing=None)
        if u is None:
            return None
        s, e = ds._indptr[u], ds._indptr[u + 1]
        fields = {k: v[s:e] for k, v in ds._attrs.items()}
        return ItemList(item_nums=ds._cols[s:e], vocabulary=ds.items, **fields)


class _Interactions:
    def __init__(self, ds):
        self._ds = ds
        self.entities = ["user", "item"]

    def matrix(self, row_entity="user", col_entity="item"):
        return _Matrix(self._ds)


class Dataset:
    "User-item interaction data: vocabularies + (user, item)-sorted interactions."

    def __init__(self, users: Vocabulary, items: Vocabulary, rows, cols, attrs: dict[str, Any]):
        self.users, self.items = users, items
        order = np.lexsort((cols, rows))
        self._rows = np.require(rows, dtype=np.int32)[order]
        self._cols = np.require(cols, dtype=np.int32)[order]
        self._attrs = {k: np.asarray(v)[order] for k, v in attrs.items()}
        # repeated (user, item) pairs are kept as separate interactions (as the reference's
        # interaction tables do); consumers that need a MATRIX sum them, as SciPy's COO -> CSR
        # conversion does on the reference's path (src/lenskit/als/_implicit.py:141-149)
        self.has_duplicates = bool(len(self._rows) > 1 and np.any(
            (self._rows[1:] == self._rows[:-1]) & (self._cols[1:] == self._cols[:-1])))
        # int32 offsets like Arrow List; int64 once the interaction count needs it
        # (src/lenskit/data/matrix.py:411-419)
        dt = np.int32 if len(self._rows) < np.iinfo(np.int32).max else np.int64
        self._indptr = np.zeros(len(users) + 1, dtype=dt)
        np.cumsum(np.bincount(self._rows, minlength=len(users)), out=self._indptr[1:])

    @property
    def user_count(self):
        return len(self.users)

    @property
    def item_count(self):
        return len(self.items)

    @property
    def interaction_count(self):
        return len(self._rows)

    def interactions(self, name: str | None = None):
        return _Interactions(self)

    def interaction_matrix(self, *, format="scipy", layout="csr", field=None):
        return _Matrix(self).scipy(field, layout=layout)

    def user_row(self, user_id) -> ItemList | None:
        return _Matrix(self).row_items(user_id)

    @classmethod
    def from_arrays(cls, user_ids, item_ids, ratings=None, *, all_item_ids=None, **attrs):
        user_ids, item_ids = np.asarray(user_ids), np.asarray(item_ids)
        users = Vocabulary(user_ids, "user")
        items = Vocabulary(item_ids if all_item_ids is None else all_item_ids, "item")
        if ratings is not None:
            attrs["rating"] = np.require(ratings, dtype=np.float32)
        return cls(users, items, users.numbers(user_ids), items.numbers(item_ids), attrs)


def from_interactions_df(df: pd.DataFrame, *, user_col="user_id", item_col="item_id",
                         rating_col="rating") -> Dataset:  # fmt: skip
    "``lenskit.data.from_interactions_df`` for (user, item[, rating]) frames."
    ucol = user_col if user_col in df else "user"
    icol = item_col if item_col in df else "item"
    ratings = df[rating_col].to_numpy() if rating_col in df else None
    return Dataset.from_arrays(df[ucol].to_numpy(), df[icol].to_numpy(), ratings)


def load_movielens_npz(path) -> Dataset:
    """
    The committed ml-latest-small fixture (tests/golden/ml_small.npz) as the reference's
    ``load_movielens`` builds it: every movies.csv id is an item
    (src/lenskit/data/sources/movielens.py:327-345).
    """
    z = np.load(path)
    return Dataset.from_arrays(z["user_id"], z["item_id"], z["rating"],
                               all_item_ids=z["all_item_ids"])  # fmt: skip


# ---------------------------------------------------------------------------------------
# MovieLens files (src/lenskit/data/sources/movielens.py:47-540)
# ---------------------------------------------------------------------------------------

_ML_NAME = r"^(ml-(?:\d+[MmKk]|latest(?:-small)?))"


class _MLSource:
    "A MovieLens distribution: a ``.zip`` (members under its top directory) or an unpacked directory."

    def __init__(self, path):
        import re
        from pathlib import Path
        from zipfile import ZipFile

        self.loc = Path(path)
        self.zip = None
        self.prefix = ""
        if self.loc.is_file() and self.loc.suffix == ".zip":
            # movielens.py:481-501: the archive's first entry is its top directory, whose name
            # gives the version
            self.zip = ZipFile(self.loc, "r")
            first = self.zip.infolist()[0].filename
            m = re.match(_ML_NAME, first)
            if not m:
                self.zip.close()
                raise RuntimeError("invalid ML zip file")
            self.version = m.group(1).lower()
            self.prefix = first if first.endswith("/") else first.split("/")[0] + "/"
            self._names = set(self.zip.namelist())
        elif self.loc.is_dir():
            m = re.match(_ML_NAME, self.loc.name)
            self.version = m.group(1).lower() if m else None
        else:
            raise RuntimeError("MovieLens data not found" if not self.loc.exists()
                               else "not a directory or zip file")

    def has(self, name: str) -> bool:
        return (self.prefix + name) in self._names if self.zip is not None \
            else (self.loc / name).exists()

    def open(self, name: str):
        return self.zip.open(self.prefix + name) if self.zip is not None \
            else open(self.loc / name, "rb")

    def close(self):
        if self.zip is not None:
            self.zip.close()

    def kind(self) -> str:
        "movielens.py:502-529: by version name, else by the files present"
        v = self.version
        if v == "ml-100k" or (v is None and self.has("u.data")):
            return "100k"
        if v in ("ml-1m", "ml-10m") or (v is None and self.has("ratings.dat")):
            return "dat"
        if v is not None or self.has("ratings.csv"):
            return "modern"
        raise RuntimeError("invalid ML directory")


def load_movielens_df(path) -> pd.DataFrame:
    """
    ``lenskit.data.load_movielens_df`` (src/lenskit/data/sources/movielens.py:455-473): the
    ratings of a MovieLens distribution (zip or directory, format detected) with columns
    ``user_id``, ``item_id``, ``rating``, ``timestamp`` and the reference's dtypes (int32 ids,
    float32 ratings: movielens.py:130-147, 283-298, 418-431).
    """
    src = _MLSource(path)
    try:
        kind = src.kind()
        if kind == "modern":
            with src.open("ratings.csv") as f:
                df = pd.read_csv(f, dtype={"userId": np.int32, "movieId": np.int32,
                                           "rating": np.float32, "timestamp": np.int64})
            df = df.rename(columns={"userId": "user_id", "movieId": "item_id"})
        elif kind == "dat":
            with src.open("ratings.dat") as f:
                df = pd.read_csv(f, sep=":", header=None, usecols=[0, 2, 4, 6],
                                 names=["user_id", "_ui", "item_id", "_ir", "rating", "_rt",
                                        "timestamp"],
                                 dtype={"user_id": np.int32, "item_id": np.int32,
                                        "rating": np.float32, "timestamp": np.int32})
        else:
            with src.open("u.data") as f:
                df = pd.read_csv(f, sep="\t", header=None,
                                 names=["user_id", "item_id", "rating", "timestamp"],
                                 dtype={"user_id": np.int32, "item_id": np.int32,
                                        "rating": np.float32, "timestamp": np.int32})
        df["timestamp"] = pd.to_datetime(df["timestamp"], unit="s")
        return df
    finally:
        src.close()


def load_movielens(path) -> Dataset:
    """
    ``lenskit.data.load_movielens`` (movielens.py:435-452) for the part of the dataset the hot
    path reads: users = the rating users, **items = every id of the movie table** (so unrated
    movies are items with empty rows: ML-25M has 62 423 items of which 59 047 are rated;
    ``MLModernLoader.dataset``, movielens.py:327-345, ``add_entities`` before the interactions,
    ids the ratings mention beyond the table inserted), interactions (user, item)-sorted with
    ``rating`` (float32) and ``timestamp``.  Titles, genres, tags and the tag genome are not read.
    """
    src = _MLSource(path)
    try:
        kind = src.kind()
        if kind == "modern":
            with src.open("movies.csv") as f:
                movie_ids = pd.read_csv(f, usecols=["movieId"],
                                        dtype={"movieId": np.int32})["movieId"].to_numpy()
        elif kind == "dat":
            with src.open("movies.dat") as f:
                movie_ids = np.array([int(line.split(b"::", 1)[0]) for line in f if line.strip()],
                                     dtype=np.int32)
        else:
            with src.open("u.item") as f:
                movie_ids = np.array([int(line.split(b"|", 1)[0]) for line in f if line.strip()],
                                     dtype=np.int32)
    finally:
        src.close()
    df = load_movielens_df(path)
    items = np.union1d(movie_ids, df["item_id"].to_numpy())  # missing="insert"
    return Dataset.from_arrays(df["user_id"].to_numpy(), df["item_id"].to_numpy(),
                               df["rating"].to_numpy(), all_item_ids=items,
                               timestamp=df["timestamp"].to_numpy())
