"""One time step of the reference's implicit driver U_Newmark_Beta (U-Newmark-beta.c:115-420), written once over an
abstract set of stage functions so that the same host algebra runs on the oracle and on the device library.
The SNES Newton-Raphson is replaced by a plain Newton iteration with a dense solve (test infrastructure only)."""
import numpy as np


def newmark_parameters(beta, gamma, dt):  # __compute_Newmark_parameters, :497-514
    return {"a1": 1 / (beta * dt * dt), "a2": 1 / (beta * dt), "a3": (1 - 2 * beta) / (2 * beta),
            "a4": gamma / (beta * dt), "a5": 1 - gamma / beta, "a6": (1 - gamma / (2 * beta)) * dt, "dt": dt}


def newmark_step(stage, ndim, bcs_list, step, nsteps, dt, gravity, beta=0.25, gamma=0.5, tol=1e-10, max_iter=12):
    """stage: object with local_search(), masks(bcs, step) -> (n2m, d2m, na), lumped_mass(), nodal_field_n(M),
    compatibility(dU, dU_dt), constitutive(), internal_forces() -- or lagrangian(dU, Un_dt, Un_dt2, M, alpha, gravity) in
    their place --, tangent(alpha_1, M) -> dense K with Dirichlet
    identity rows, roll(), update_kinetics(dU, Un_dt, dU_dt, dU_dt2).  Returns (dU, residual history)."""
    a = newmark_parameters(beta, gamma, dt)
    stage.local_search()                                           # :197
    n2m, d2m, na = stage.masks(bcs_list, step)                     # :205-209
    ntot = na * ndim
    free = d2m != -1
    M = stage.lumped_mass()                                        # :223
    Un_dt, Un_dt2 = stage.nodal_field_n(M)                         # :241
    alpha = [a["a1"], a["a2"], a["a3"], a["a4"], a["a5"], a["a6"]]
    if hasattr(stage, "form_initial_guess"):                       # device versions of the per-dof updates (a21)
        dU = stage.form_initial_guess(Un_dt, Un_dt2, dt, bcs_list, step)
    else:
        dU = dt * Un_dt + 0.5 * dt * dt * Un_dt2                   # __form_initial_guess, explicit trial :893-901
        for b in bcs_list:                                         # :909-950
            for node in b["nodes"]:
                m = n2m[node]
                if m == -1:
                    continue
                for k in range(b["dim"]):
                    if b["dir"][k, step] == 1:
                        dU[m * ndim + k] = b["value"][k, step]
    bvec = np.tile(np.asarray(gravity, dtype=np.float64), na)
    history = []
    for it in range(max_iter):
        if hasattr(stage, "kinetic_increments"):
            dU_dt, _ = stage.kinetic_increments(dU, Un_dt, Un_dt2, alpha)
        else:
            dU_dt = a["a4"] * dU + (a["a5"] - 1) * Un_dt + a["a6"] * Un_dt2    # :1836-1856
        if hasattr(stage, "lagrangian"):                                        # __lagrangian_evaluation as one call
            R = stage.lagrangian(dU, Un_dt, Un_dt2, M, alpha, gravity)
        else:
            stage.compatibility(dU, dU_dt)                                          # :1021
            stage.constitutive()                                                    # :1026
            R = stage.internal_forces()                                             # :1028 (Dirichlet dofs skipped)
            if hasattr(stage, "inertial_forces"):
                R = stage.inertial_forces(R, M, dU, Un_dt, Un_dt2, alpha, gravity)
            else:
                R[free] += (M * (a["a1"] * dU - a["a2"] * Un_dt - a["a3"] * Un_dt2 - bvec))[free]   # :1519-1557
        res = float(np.linalg.norm(R[free]))
        history.append(res)
        if res <= tol * max(1.0, history[0]):
            break
        K = stage.tangent(a["a1"], M)                                           # :1646-1830
        rhs = -R
        rhs[~free] = 0.0
        dU = dU + np.linalg.solve(K, rhs)
    if hasattr(stage, "kinetic_increments"):
        dU_dt, dU_dt2 = stage.kinetic_increments(dU, Un_dt, Un_dt2, alpha)
    else:
        dU_dt = a["a4"] * dU + (a["a5"] - 1) * Un_dt + a["a6"] * Un_dt2         # :1859-1906
        dU_dt2 = a["a1"] * dU - a["a2"] * Un_dt - (a["a3"] + 1) * Un_dt2
    stage.roll()                                                                # :393
    stage.update_kinetics(dU, Un_dt, dU_dt, dU_dt2)                             # :396, alpha_blend = 1 (:148)
    return dU, history
