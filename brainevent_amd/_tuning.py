"""Per-architecture tuning of the scatter routes, persisted as JSON — the counterpart of the reference's per-GPU store of its
hybrid kernel's thresholds (``brainevent/_csr/hybrid_config.py:77-88`` the record, ``:229-255`` resolution order,
``:256-295`` the per-device JSON store, ``brainevent/_csr/initialize.py:182-186`` the explicit tuner).  What is tuned here
is what this design has instead of ``tpr_threshold`` / ``task_nnz``: from how many entries per (row, slice) block the
planned layout beats the binned route, from how many stored entries a layout pays at all, and the task size of the binned
route's pass B.  The defaults are the values measured on gfx950 (MI355X); nothing is measured implicitly —
``tune_scatter_routes`` is the explicit call, as in the reference.

Resolution order (same as the reference's): environment variable ``BRAINEVENT_AMD_TUNING`` (a JSON object), then the entry of
the current device kind (``gcnArchName`` without feature flags, e.g. ``gfx950``) in the JSON store
(``BRAINEVENT_AMD_TUNING_FILE`` or ``~/.cache/brainevent_amd/scatter_tuning.json``), then the defaults.  A corrupt or partial
file never breaks anything: the defaults are used."""
import dataclasses
import json
import os
from pathlib import Path
from typing import Mapping, Optional, Sequence

__all__ = ['ScatterTuning', 'DEFAULT_SCATTER_TUNING', 'get_scatter_tuning', 'save_scatter_tuning', 'apply_scatter_tuning',
           'current_device_kind', 'tune_scatter_routes', 'ensure_resolved']

_ENV_OVERRIDE = 'BRAINEVENT_AMD_TUNING'
_ENV_FILE = 'BRAINEVENT_AMD_TUNING_FILE'


@dataclasses.dataclass(frozen=True)
class ScatterTuning:
    plan_min_nnz: int = 1 << 15             # stored entries below which no per-matrix layout is built (direct route)
    plan_min_segment: int = 8               # entries per (row, slice) block from which the planned layout beats the binned route
    plan_min_segment_homo: int = 10         # ... for one shared weight
    plan_min_segment_no_binned: int = 8     # ... and from which it beats the direct route where the binned route does not apply
    binned_task_groups: int = 256           # pass B of the binned route: groups of four entries per task (round 4: 1024 -> 256, >= 4 rows)
    binned_min_tasks: int = 2048            # ... and the tasks a step is cut into at least
    benchmark_records: tuple = ()           # what the tuner measured (kept with the entry, not interpreted)

    def validated(self) -> 'ScatterTuning':
        for f in dataclasses.fields(self):
            if f.name == 'benchmark_records':
                continue
            v = getattr(self, f.name)
            if not isinstance(v, int) or isinstance(v, bool) or v < 1 or v >= 1 << 31:
                raise ValueError(f"ScatterTuning.{f.name} must be an integer in [1, 2^31), got {v!r}.")
        return self


DEFAULT_SCATTER_TUNING = ScatterTuning()


def _from_mapping(m: Mapping) -> ScatterTuning:
    names = {f.name for f in dataclasses.fields(ScatterTuning)} - {'benchmark_records'}
    kw = {k: int(v) for k, v in m.items() if k in names}
    rec = m.get('benchmark_records') or ()
    return ScatterTuning(**kw, benchmark_records=tuple(dict(r) for r in rec)).validated()


def _to_mapping(t: ScatterTuning) -> dict:
    d = {f.name: getattr(t, f.name) for f in dataclasses.fields(t) if f.name != 'benchmark_records'}
    if t.benchmark_records:
        d['benchmark_records'] = [dict(r) for r in t.benchmark_records]
    return d


def _store_path() -> Path:
    p = os.environ.get(_ENV_FILE)
    return Path(p) if p else Path.home() / '.cache' / 'brainevent_amd' / 'scatter_tuning.json'


def current_device_kind() -> Optional[str]:
    """``gcnArchName`` of the current HIP device without its feature flags (``gfx950``); None without a device."""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
        name = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), 'gcnArchName', '') or ''
        return name.split(':')[0] or None
    except Exception:          # noqa: BLE001 - a machine without a working runtime simply has no device kind
        return None


_warned = set()


def _warn_once(key: str, msg: str) -> None:
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn(msg)


def _env_override() -> Optional[ScatterTuning]:
    """The tuning of ``BRAINEVENT_AMD_TUNING``, ``None`` when unset.  A malformed value is reported once and ignored (the
    defaults): a bad override must never break the kernels — neither at import nor at any later resolution."""
    raw = os.environ.get(_ENV_OVERRIDE)
    if not raw:
        return None
    try:
        return _from_mapping(json.loads(raw))
    except (ValueError, TypeError, KeyError, AttributeError) as e:
        _warn_once('env:' + raw, f"brainevent_amd: {_ENV_OVERRIDE} ignored ({e!r}); using the built-in defaults.")
        return DEFAULT_SCATTER_TUNING


_resolved = {}          # device kind (or None) -> ScatterTuning


def get_scatter_tuning(resolve_device: bool = True) -> ScatterTuning:
    """The tuning of this process for the current device kind (memoised per kind; never measures anything).

    ``resolve_device=False`` is what the package import uses: the environment override or the defaults, WITHOUT asking the
    runtime which device is current — that query initialises the GPU (every torchrun rank importing before ``set_device``
    would open a context on GPU 0, fork-after-import breaks, an exec-before-GPU re-launch pattern stops being one).  The
    per-architecture entry of the store is resolved at the first route choice / binned workspace instead
    (:func:`ensure_resolved`), after the caller has picked its device."""
    env = _env_override()
    if env is not None:
        return env
    if not resolve_device:
        return DEFAULT_SCATTER_TUNING
    path = _store_path()
    if not path.exists():
        return DEFAULT_SCATTER_TUNING
    kind = current_device_kind()
    key = (str(path), kind)
    if key not in _resolved:
        t = DEFAULT_SCATTER_TUNING
        try:
            entry = json.loads(path.read_text(encoding='utf-8')).get(kind)
            if entry is not None:
                t = _from_mapping(entry)
        except (OSError, ValueError, KeyError, TypeError, AttributeError):
            pass               # a corrupt / partial file must never break the kernels: defaults
        _resolved[key] = t
    return _resolved[key]


def _cache_clear() -> None:
    _resolved.clear()
    _applied[0] = None


get_scatter_tuning.cache_clear = _cache_clear       # (the name the memoised function of rounds 1-3 offered)

_applied = [None]        # (store path, device kind) whose entry the module constants currently reflect


def ensure_resolved() -> None:
    """Called where a device is certainly in use (route choice, binned workspace): resolve the store entry of the CURRENT
    device kind once and apply it; later calls are a dictionary lookup.  Without a store, or with an environment override (applied
    at import already), nothing touches the runtime."""
    if os.environ.get(_ENV_OVERRIDE):
        return
    path = _store_path()
    if not path.exists():
        return
    if _applied[0] == 'explicit':      # the caller applied a tuning of its own (apply_scatter_tuning): that stands
        return
    key = (str(path), current_device_kind())
    if _applied[0] != key:
        apply_scatter_tuning(get_scatter_tuning(), _resolved_key=key)


def save_scatter_tuning(tuning: ScatterTuning, device_kind: Optional[str] = None,
                        benchmark_records: Optional[Sequence[Mapping]] = None) -> Path:
    """Persist ``tuning`` for ``device_kind`` (default: the current device) — the other devices' entries of the store are
    kept — re-resolve this process's tuning and apply it.  Returns the store's path."""
    tuning = tuning.validated()
    device_kind = device_kind or current_device_kind()
    if not device_kind:
        raise RuntimeError("cannot determine the device kind; pass device_kind explicitly")
    if benchmark_records is not None:
        tuning = dataclasses.replace(tuning, benchmark_records=tuple(dict(r) for r in benchmark_records))
    path = _store_path()
    path.parent.mkdir(parents=True, exist_ok=True)
    store = {}
    if path.exists():
        try:
            store = json.loads(path.read_text(encoding='utf-8'))
            if not isinstance(store, dict):
                store = {}
        except ValueError:
            store = {}
    store[device_kind] = _to_mapping(tuning)
    tmp = path.with_suffix(path.suffix + f'.{os.getpid()}.tmp')
    tmp.write_text(json.dumps(store, indent=2, sort_keys=True), encoding='utf-8')
    os.replace(tmp, path)
    get_scatter_tuning.cache_clear()
    apply_scatter_tuning()
    return path


_pushed = [None]


def push_to_library(tuning: Optional[ScatterTuning] = None) -> None:
    """Hand the binned route's task size to the loaded library (``be_binned_set_tuning``); a no-op when nothing changed."""
    if tuning is None:
        ensure_resolved()
        from . import _csr as C
        tuning = getattr(C, '_TUNING', None)
    t = tuning or get_scatter_tuning()
    key = (t.binned_task_groups, t.binned_min_tasks)
    if _pushed[0] == key:
        return
    import ctypes
    from ._lib import fn, check
    check(fn('be_binned_set_tuning', ctypes.c_int, [ctypes.c_int, ctypes.c_int])(*key), 'be_binned_set_tuning')
    _pushed[0] = key


def apply_scatter_tuning(tuning: Optional[ScatterTuning] = None, _resolved_key=None) -> ScatterTuning:
    """Make ``tuning`` (default: the resolved one) what ``choose_scatter_route`` uses: the module constants of
    ``brainevent_amd._csr`` — and, once the library is loaded, the task size of the binned route.  A tuning applied by the caller
    stands until ``get_scatter_tuning.cache_clear()``; the lazy per-device resolution does not replace it."""
    from . import _csr as C
    t = (tuning or get_scatter_tuning()).validated()
    _applied[0] = _resolved_key if _resolved_key is not None else ('explicit' if tuning is not None else _applied[0])
    C.PLAN_MIN_NNZ = t.plan_min_nnz
    C.PLAN_MIN_SEGMENT = t.plan_min_segment
    C.PLAN_MIN_SEGMENT_HOMO = t.plan_min_segment_homo
    C.PLAN_MIN_SEGMENT_NO_BINNED = t.plan_min_segment_no_binned
    C._TUNING = t
    from . import _lib
    if getattr(_lib, '_lib', None) is not None:          # (never loads the library by itself)
        push_to_library(t)
    return t


def tune_scatter_routes(*, sizes=(500_000, 1_000_000, 1_500_000), conn: int = 1000, fire: float = 0.01, steps: int = 30,
                        save: bool = True, verbose: bool = False) -> ScatterTuning:
    """Measure, on the current device, the entries per (row, slice) block at which the planned layout and the binned route
    cross (``FixedNumPerPre`` with ``conn`` entries per row over ``sizes`` outputs, ``fire`` of the rows active; one shared
    weight and per-entry weights) and return — ``save=True``: persist — a tuning with those thresholds.  The explicit
    counterpart of the reference's ``init_csr_config`` (``brainevent/_csr/initialize.py:182-186``); a minute on an MI355X."""
    import time
    import torch
    from . import _csr as C
    from ._event import BinaryArray
    from ._fcn import FixedNumPerPre
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    base = get_scatter_tuning()
    records, cross = [], {True: [], False: []}
    saved = (C.PLAN_MIN_SEGMENT, C.PLAN_MIN_SEGMENT_HOMO)
    try:
        for n in sizes:
            idx = torch.randint(0, n, (n, conn), dtype=torch.int32, device=dev, generator=g)
            spikes = [torch.rand(n, device=dev, generator=g) < fire for _ in range(4)]
            for homo in (True, False):
                w = torch.ones(1, device=dev) if homo else torch.rand((n, conn), device=dev, generator=g)
                t_us, per_block = {}, None
                for route, seg_min in (('plan', 1), ('binned', 1 << 30)):
                    C.PLAN_MIN_SEGMENT = C.PLAN_MIN_SEGMENT_HOMO = seg_min
                    m = FixedNumPerPre((w, idx), shape=(n, n), check_indices=False).prepare()
                    ws = m.buffers.get('scatter_plan')
                    if isinstance(ws, C.ScatterPlan):
                        per_block = conn / ws.n_slices
                    for i in range(3):
                        BinaryArray(spikes[i % 4]) @ m
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(steps):
                        BinaryArray(spikes[i % 4]) @ m
                    torch.cuda.synchronize()
                    t_us[route if isinstance(ws, C.ScatterPlan) == (route == 'plan') else type(ws).__name__] = \
                        (time.perf_counter() - t0) / steps * 1e6
                    del m, ws
                    torch.cuda.empty_cache()
                rec = {'n': int(n), 'conn': int(conn), 'homo': bool(homo), 'entries_per_block': per_block,
                       **{f'{k}_us': round(v, 1) for k, v in t_us.items()}}
                records.append(rec)
                if verbose:
                    print(rec, flush=True)
                if per_block and 'plan' in t_us and 'binned' in t_us:
                    cross[homo].append((per_block, t_us['plan'] <= t_us['binned']))
                del w
            del idx, spikes
            torch.cuda.empty_cache()
    finally:
        C.PLAN_MIN_SEGMENT, C.PLAN_MIN_SEGMENT_HOMO = saved

    def threshold(points, default):
        wins = [pb for pb, plan_wins in points if plan_wins]
        loses = [pb for pb, plan_wins in points if not plan_wins]
        if wins and loses and min(wins) > max(loses):
            return max(1, int(round((min(wins) + max(loses)) / 2)))
        if wins and not loses:
            return max(1, int(min(wins)))            # the plan won everywhere measured: at least down to here
        return default                               # no clean crossing in the measured range: keep what was there
    out = dataclasses.replace(base, plan_min_segment=threshold(cross[False], base.plan_min_segment),
                              plan_min_segment_homo=threshold(cross[True], base.plan_min_segment_homo),
                              benchmark_records=tuple(records))
    if save:
        save_scatter_tuning(out)
    return out
