"""Name -> operator table (reference ``brainevent/_registry.py:39-108``)."""
from typing import Dict, List, Set

_REGISTRY: Dict[str, 'OpKernel'] = {}


def register_primitive(name: str, primitive) -> None:
    if name in _REGISTRY and _REGISTRY[name] is not primitive:
        raise ValueError(f"primitive {name!r} is already registered.")
    _REGISTRY[name] = primitive


def get_registry() -> Dict[str, 'OpKernel']:
    return dict(_REGISTRY)


def get_primitives_by_tags(tags: Set[str]) -> Dict[str, 'OpKernel']:
    tags = set(tags)
    return {k: v for k, v in _REGISTRY.items() if tags.issubset(v.tags)}


def get_all_primitive_names() -> List[str]:
    return sorted(_REGISTRY)
