"""Host-side control logic of the ALM loop: penalty / scaling policy, lazy KKT validation,
running history.  These decide *which* device work runs on which iteration and when the loop
stops, so they follow the reference decision for decision:

    AdjustAdmmParam           dot_surface_socp/utils/admm_tools.py:19-171
    ConditionValidator        dot_surface_socp/utils/condition_validator.py:194-331 (+ wrappers :105-191)
    AdaptiveValidator         dot_surface_socp/utils/condition_validator_wrapper.py:9-151
    RunningHistory            dot_surface_socp/utils/admm_tools.py:174-251, 399-442, 505-562

They are plain Python (the reference's are too) and never touch array data: every number they
see is a scalar that came back through the C ABI.
"""
from __future__ import annotations

import logging
import math
import time
from contextlib import contextmanager
from typing import Callable, List, Optional, Sequence

import numpy as np

logger = logging.getLogger("dots_socp_amd")


# --------------------------------------------------------------------------- penalty policy
class AdjustAdmmParam:
    """When and by how much the penalty r changes; when z is re-scaled."""

    # (threshold on max(gap, 1/gap), factor): admm_tools.py:79-90
    FACTOR_STEPS = ((50.0, 2.00), (35.0, 1.75), (20.0, 1.60), (10.0, 1.40), (5.0, 1.35), (3.0, 1.32),
                    (2.5, 1.28), (2.0, 1.26), (1.5, 1.20), (1.2, 1.10))
    # (iteration bound, minimum gap since the last adjustment): admm_tools.py:43-48
    SCHEDULE = ((20, 3), (50, 7), (100, 11), (200, 17), (500, 31))
    LATE_GAP = 43

    def __init__(self):
        self.last_it = -1
        self._sigma_upper_bound = 10.0 ** 3
        self._sigma_lower_bound = 10.0 ** (-3)
        self._scale_times_matrix = 0

    def peek_adjust(self, current_it: int) -> bool:
        passed = current_it - self.last_it
        for bound, gap in self.SCHEDULE:
            if current_it < bound and passed >= gap:
                return True
        return passed >= self.LATE_GAP

    def is_to_adjust(self, current_it: int) -> bool:
        if self.peek_adjust(current_it):
            self.last_it = current_it
            return True
        return False

    @classmethod
    def adjust_factor(cls, prim_dual_gap: float) -> float:
        prim_win = prim_dual_gap < 1.0
        gap = 1.0 / prim_dual_gap if prim_win else prim_dual_gap
        factor = 1.0
        for bound, val in cls.FACTOR_STEPS:
            if gap > bound:
                factor = val
                break
        return 1.0 / factor if prim_win else factor

    def get_updated_value(self, sigma: float, prim_dual_gap: float) -> float:
        return max(min(sigma * self.adjust_factor(prim_dual_gap), self._sigma_upper_bound), self._sigma_lower_bound)

    @staticmethod
    def is_to_scale(current_it: int) -> bool:
        return current_it == 10 or current_it == 50 or current_it % 100 == 50

    def is_to_scale_matrix(self, current_it, current_kkt, min_it=100, max_scale_times=1, tol=5e-3) -> bool:
        # Python's builtin max on purpose: with NaN entries (conditions not evaluated on that
        # iteration) its result depends on the order, exactly as in the reference (:107-110).
        if current_it >= min_it and self._scale_times_matrix < max_scale_times and max(list(current_kkt)) < tol:
            self._scale_times_matrix += 1
            return True
        return False

    @staticmethod
    def compute_scale_factor(prim_norm, dual_norm):
        return float(np.max(prim_norm)), float(np.max(dual_norm))


# --------------------------------------------------------------------------- lazy KKT validation
def max_of_list_with_none(values):
    vals = [v for v in values if v is not None]
    return max(vals) if vals else None


class ErrorCondition:
    """One KKT condition: callable returning [value, value_at_unit_scale]; passes if value < tol."""

    def __init__(self, fn: Callable[[], Sequence[Optional[float]]], tol: float, name: str):
        self.fn, self.tol, self.name = fn, tol, name
        self.last = [None, None]

    def __call__(self) -> bool:
        try:
            self.last = list(self.fn())
            return bool(self.last[0] < self.tol)
        except (ArithmeticError, ValueError, TypeError) as exc:
            # the reference swallows errors of the evaluation and reports inf (condition_validator.py:140-150);
            # a device / state error of the library (HipLibraryError, a RuntimeError) is NOT "not converged": it propagates
            logger.error("Error in condition %r: %s", self.name, exc)
            self.last = [float("inf"), float("inf")]
            return False

    def take(self):
        out, self.last = self.last, [None, None]
        return out


class ConditionValidator:
    """Circular queue over the conditions with early exit on the first failure."""

    def __init__(self, conditions: List[ErrorCondition], queue_order: Sequence[int]):
        if sorted(queue_order) != list(range(len(conditions))):
            raise ValueError("queue_order must be a permutation of the condition indices")
        self.by_id = conditions                  # original order, used by the collector
        self.queue = list(queue_order)           # queue slot -> condition id
        self.size = len(conditions)
        self.front = 0

    def get_num_conditions(self):
        return self.size

    def validate(self, required_conditions=None):
        required = list(required_conditions or [])
        checked, failing = [], []

        def run(cond_id):
            if cond_id in checked:
                return True
            checked.append(cond_id)
            ok = self.by_id[cond_id]()
            if not ok:
                failing.append(cond_id)
            return ok

        required_ok = [run(c) for c in required if c < self.size]
        all_passed = False
        if all(required_ok):
            if len(checked) >= self.size:
                all_passed = True
            else:
                start = self.front
                while len(checked) < self.size:
                    cond_id = self.queue[self.front % self.size]
                    if cond_id not in checked and not run(cond_id):
                        break
                    self.front = (self.front + 1) % self.size
                    if self.front == start:
                        all_passed = True
                        break
        info = {
            "all_passed": all_passed,
            "num_conditions_passed": len(checked),
            "failing_conditions": failing,
            "early_termination": (not all_passed) and len(checked) < self.size,
        }
        return all_passed, info

    def collect(self):
        """(values with scale, values at unit scale) in original order; resets the stored values."""
        pairs = [c.take() for c in self.by_id]
        return [p[0] for p in pairs], [p[1] for p in pairs]


class AdaptiveValidator:
    """Validate only every `interval` calls; the interval follows error / tolerance."""

    def __init__(self, validator: ConditionValidator, min_interval=1, max_interval=37):
        self.validator = validator
        self.min_interval, self.max_interval = min_interval, max_interval
        self.current_interval = 1
        self.iteration_counter = 0

    def get_num_conditions(self):
        return self.validator.get_num_conditions()

    def reset_counter(self):
        self.iteration_counter = 0

    def set_error_and_tolerance(self, error, tolerance):
        if isinstance(error, float) and isinstance(tolerance, float):      # the solver's call: plain floats, no numpy round trips
            ratio = error / max(tolerance, 1e-10)
            if ratio <= 1.0:
                self.current_interval = self.min_interval
                return
            log_ratio = math.log10(ratio) if ratio == ratio and ratio != math.inf else ratio      # nan, inf pass through as in numpy
        else:
            ratio = float(np.max(np.atleast_1d(error) / np.maximum(np.atleast_1d(tolerance), 1e-10)))
            if ratio <= 1.0:
                self.current_interval = self.min_interval
                return
            log_ratio = np.log10(ratio)
        if log_ratio > 1.0:
            self.current_interval = self.max_interval
        else:
            self.current_interval = max(self.min_interval,
                                        int(self.min_interval + log_ratio * (self.max_interval - self.min_interval)))

    def will_validate_next(self) -> bool:
        return (self.iteration_counter % self.current_interval) == 0

    def validate(self, required_conditions=None):
        due = self.will_validate_next()
        self.iteration_counter += 1
        if due or required_conditions:
            return self.validator.validate(required_conditions)
        return False, {}

    def collect(self):
        return self.validator.collect()


# --------------------------------------------------------------------------- running history
KKT_LABELS = [
    "SOC & Org : Primal Feasibility (q)",
    "SOC       : Primal Feasibility (z)",
    "SOC & Org : Dual Feasibility (alpha)",
    "SOC       : Dual Feasibility (beta)",
    "      Org : ||rho - Pi+(rho + Fq)||",
    "      Org : ||m - rho o B||",
    "      Org : ||cong. rho - lambda_c||",
]
KKT_SHORT_LABELS = ["Prim(phi, q)", "Prim(q, z)", "Dual(alpha)", "Dual(beta)", "Comp(rho, f(q))", "Comp(m, rho o B)",
                    "Comp(rho, cong.)"]


class RunningHistory:
    """KKT / time / iteration records with the attribute names the reference's interface reads
    (``history``, ``kkt_errors``, ``kkt_iteration``, ``running_time``, ``steps_time``)."""

    def __init__(self, max_record_numbers: int, kkt_labels=None, name="SOCP", kkt_short_labels=None):
        self.kkt_labels = list(kkt_labels or KKT_LABELS)
        self.kkt_short_labels = list(kkt_short_labels or KKT_SHORT_LABELS)
        if len(self.kkt_short_labels) != len(self.kkt_labels):
            raise ValueError("kkt_short_labels must have the same length as kkt_labels")
        self.name = name
        self.kkt_entry_num = len(self.kkt_labels)
        self._max_num = int(max_record_numbers)
        self._kkt_num = 0
        self.kkt_errors = np.full((self._max_num, self.kkt_entry_num), np.inf)
        self.kkt_iteration = np.full(self._max_num, np.inf)
        self.kkt_time = np.full(self._max_num, np.inf)
        self.running_time = np.inf
        self.last_record_it = -1
        self.steps_time: dict = {}
        self.history: dict = {}
        self.solver_stats: dict = {}
        self._t0 = np.inf

    def start(self):
        self._t0 = time.perf_counter()

    def get_running_time(self):
        return time.perf_counter() - self._t0

    def end(self):
        self.running_time = time.perf_counter() - self._t0
        n = self._kkt_num
        self.kkt_errors = self.kkt_errors[:n]
        self.kkt_iteration = self.kkt_iteration[:n]
        self.kkt_time = self.kkt_time[:n]
        for key in self.history:
            self.history[key] = self.history[key][:n]

    @contextmanager
    def timer(self, tag):
        t0 = time.perf_counter()
        yield
        self.add_time(tag, time.perf_counter() - t0)

    def add_time(self, tag, seconds):
        self.steps_time[tag] = self.steps_time.get(tag, 0.0) + seconds

    steps_time_note = None      # how steps_time was obtained when it is not a plain sum of timers (SampledStepTimers)

    def record(self, current_it=None, kkt_errors=None, history=None):
        if kkt_errors is None or current_it is None:
            raise ValueError("Argument `kkt_errors` or `current_it` must be provided.")
        if current_it < self.last_record_it:
            raise ValueError(f"Current iteration {current_it} is smaller than last recorded iteration {self.last_record_it}.")
        if current_it == self.last_record_it:
            self._kkt_num -= 1          # same iteration again: overwrite the last row
        if self._kkt_num >= self._max_num:
            raise ValueError("There is no redundant space to store the running history.")
        self.last_record_it = current_it
        self.kkt_errors[self._kkt_num] = [np.nan if e is None else e for e in kkt_errors]
        self.kkt_iteration[self._kkt_num] = current_it
        self.kkt_time[self._kkt_num] = time.perf_counter() - self._t0
        if history is not None:
            for key, val in history.items():
                if key not in self.history:
                    self.history[key] = np.full_like(self.kkt_iteration, np.inf)
                self.history[key][self._kkt_num] = val
        self._kkt_num += 1

    def get_current_kkt_errors(self):
        if self._kkt_num == 0:
            return np.full(self.kkt_entry_num, np.inf)
        return self.kkt_errors[self._kkt_num - 1]

    # ---- end-of-run reports (same lines the reference prints; replication/log2table.py:99-106 scrapes them)
    @staticmethod
    def _sep(title):
        return f"---- {title} ".ljust(42, "-")

    def print_end_history(self):
        width = max(len(s) for s in self.kkt_labels)
        lines = [self._sep("The kkt errors at end")]
        lines += [f"{lab:<{width}}: {err:>6.2e}" for lab, err in zip(self.kkt_labels, self.kkt_errors[-1])]
        if self.history:
            lines.append(self._sep("Other history at end"))
            lines += [f"{key}: {val[-1]:.6e}" for key, val in self.history.items()]
        logging.getLogger().info("\n".join(lines))

    def print_steps_time(self, tag_tips="Time of each step", tag_step_time="Time of steps", tag_total_time="Total Time",
                         tag_total_iteration="Total Iteration"):
        total_time, total_it = self.running_time, self.kkt_iteration[-1]
        labels = list(self.steps_time)
        width = max(len(s) for s in labels + [tag_step_time, tag_total_time, tag_total_iteration])
        lines = [self._sep(tag_tips)]
        for lab in labels:
            t = self.steps_time[lab]
            lines.append(f"{lab:<{width}}: {t:>7.2f} sec ({100.0 * t / total_time:5.2f}%) ({100.0 * t / max(total_it, 1):<5.2f} sec/100-iterations)")
        s = sum(self.steps_time.values())
        lines.append("-" * 42)
        lines.append(f"{tag_step_time.ljust(width)}: {s:>7.2f} sec ({100.0 * s / total_time:5.2f}%) ({100.0 * s / max(total_it, 1):<5.2f} sec/100-iterations)")
        lines.append(f"{tag_total_time.ljust(width)}: {total_time:>7.2f} sec ({100.0:5.2f}%)")
        lines.append(f"{tag_total_iteration.ljust(width)}: {total_it:>7.0f} iterations")
        if self.steps_time_note:
            lines.append(f"({self.steps_time_note})")
        logging.getLogger().info("\n".join(lines))

    def show_kkt_errors(self, filename=None, is_show_when_save=False, x_axis="iteration", title=None, x_label=None, y_label=None):
        import matplotlib

        if filename is not None and not is_show_when_save:
            matplotlib.use("Agg")
        from matplotlib import pyplot as plt

        if x_axis == "iteration":
            x, xl = self.kkt_iteration, "Iteration numbers"
        elif x_axis == "time":
            x, xl = self.kkt_time, "Iteration time [seconds]"
        else:
            raise ValueError(f"x_axis {x_axis} is not supported.")
        fig = plt.figure()
        for n in range(self.kkt_entry_num):
            y = np.where(self.kkt_errors[:, n] < 1e-10, 0.0, self.kkt_errors[:, n])
            plt.semilogy(x, y, label=self.kkt_short_labels[n])
        plt.title(title if isinstance(title, str) else self.name)
        plt.xlabel(x_label if isinstance(x_label, str) else xl)
        plt.ylabel(y_label if isinstance(y_label, str) else "Karush-Kuhn-Tucker errors")
        plt.legend()
        if isinstance(filename, str):
            fig.savefig(filename, bbox_inches="tight")
        plt.close(fig)


def safe_rescale_ratio(prim_gap: float, kkt_row) -> float:
    """prim_gap * sqrt(kkt[1] / kkt[0]) as at solver_socp.py:664, NaN-transparent."""
    num, den = float(kkt_row[1]), float(kkt_row[0])
    q = num / den if den != 0.0 else (float("inf") if num > 0 else float("nan"))
    return prim_gap * math.sqrt(q) if q >= 0.0 else float("nan")


class SampledStepTimers:
    """``RunningHistory.steps_time`` (the reference's per-step timers, utils/admm_tools.py:244-251, printed as "Time of steps"
    at :505-540 and scraped into the paper's Time[s] column by replication/log2table.py:99-106) for a loop that never waits for
    the device: the phases of SAMPLED iterations are bracketed by events on the stream (DOTS_STEP_TIMED; the library takes the
    measured cost of an event off every phase) and collected at the next read-back.  Iterations come in kinds (e.g. "quiet":
    nothing read back, z_mid not stored; "read-back") that cost differently, and the first iterations of a run are not typical
    (cold caches and kernels, a dense KKT / penalty schedule).  So each kind has two strata:

        the first ``first`` iterations of the kind     counted EXACTLY (every one of them is timed)
        the iterations after them                       (median of the every-``every``-th samples)  x  (their number)

    and the time of a step is the sum over kinds and strata: the device time of ALL iterations' steps, not of the sampled
    ones.  Until the second stratum has a sample of its own its iterations are priced at the LAST timed iteration of the
    kind.  ``every = 1``: every iteration is timed and the estimate is the plain sum."""

    def __init__(self, history, first=4, every=8):
        self.history, self.first, self.every = history, int(first), max(1, int(every))
        self.n = {}          # kind -> iterations so far
        self.values = {}     # kind -> tag -> [seconds of each timed iteration, oldest first]
        self.partial = {}    # kind -> tag -> seconds of the timed iteration whose records are still arriving
        self.tags = []

    def begin(self, kind):
        """Count one iteration of ``kind``; True if it is to be sampled."""
        n = self.n[kind] = self.n.get(kind, 0) + 1
        return n <= self.first or n % self.every == 0

    def add(self, kind, tag, seconds, samples=1):
        """``samples``: 1 on the record that completes a timed iteration (a time slab reports its stages one by one: the record of
        its last stage carries 1, the others 0)."""
        if tag not in self.tags:
            self.tags.append(tag)
        part = self.partial.setdefault(kind, {})
        part[tag] = part.get(tag, 0.0) + seconds
        if samples:
            self.values.setdefault(kind, {}).setdefault(tag, []).append(part.pop(tag))

    def publish(self):
        hist = self.history
        sampled = total = 0
        for tag in self.tags:
            est = 0.0
            for kind, n in self.n.items():
                vals = self.values.get(kind, {}).get(tag)
                if not vals:
                    continue
                head, tail = vals[:self.first], vals[self.first:]
                est += sum(head)
                rest = n - len(head)
                if rest <= 0:
                    continue
                if len(vals) >= n:                   # every iteration was timed: the plain sum
                    est += sum(tail)
                elif tail:
                    est += float(np.median(tail)) * rest
                else:                                # no periodic sample yet: the latest timed iteration stands in
                    est += head[-1] * rest
            hist.steps_time[tag] = est
        for kind, n in self.n.items():
            total += n
            sampled += max([len(v) for v in self.values.get(kind, {}).values()] or [0])
        if sampled < total:
            hist.steps_time_note = (f"device time of all {total} iterations, estimated per kind of iteration "
                                    f"({', '.join(f'{n} {k}' for k, n in self.n.items())}) from {sampled} timed ones (the first {self.first} of a kind "
                                    f"counted exactly, the others at the median of every {self.every}-th): events on the stream, read after the fact, "
                                    "no host wait in the loop; a launch shared by the right-hand side and the projection is booked half to each")
        else:
            hist.steps_time_note = None
