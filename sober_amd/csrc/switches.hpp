#pragma once
// A/B and test switches of the environment, read ONCE when the library is loaded (not per level, not per call):
// sober_reload_switches() re-reads them (the tests that flip one inside a process call it through sober_amd._native).
namespace sober {
struct Switches {
    bool level_two_launches;    // SOBER_LEVEL_TWO_LAUNCHES: the leftover positions' sums as a launch of their own
    bool tani_no_queue;         // SOBER_TANI_NO_QUEUE: fingerprint levels sized by the host after a synchronisation
    bool car_force_giveup;      // SOBER_CAR_FORCE_GIVEUP: launches that wait for partner workgroups give up at once
    bool car_unfused;           // SOBER_CAR_UNFUSED: bidiagonalisation and Phi as two launches
    bool car_exact_ratio;       // SOBER_CAR_EXACT_RATIO: the pivots' ratio test without its screened fast path (what the tests compare it with)
    bool level_no_classes;      // SOBER_LEVEL_NO_CLASSES: every level's set sums evaluated (no levels derived from class sums, level_class.hip)
};
const Switches& switches();     // (misc.hip)
}

