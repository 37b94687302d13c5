"""Mirror of reference src/utils/common.py:54-70 (ModelOutput container)."""
from collections import OrderedDict
from dataclasses import fields
from typing import Any


class ModelOutput(OrderedDict):
    def keys(self) -> Any:
        for field in fields(self):  # type: ignore
            yield field.name

    def __getitem__(self, key: Any) -> Any:
        return getattr(self, key)

    def __iter__(self) -> Any:
        yield from self.keys()

    def values(self) -> Any:
        for field in fields(self):  # type: ignore
            yield getattr(self, field.name)

    def items(self) -> Any:
        for field in fields(self):  # type: ignore
            yield field.name, getattr(self, field.name)
