// dump_reference.rs -- the pin a Magnetite maintainer can run (UNCOMPILED here: this image has no cargo / rustc).
//
// PARITY of this repository's oracle (oracle/magnetite_oracle.c) is UNPINNED against reference outputs: the reference
// ships no tests or vectors and cannot be built in the build environment.  This file closes that gap on any machine
// with cargo.  It is a `#[cfg(test)]` child module of the reference's `src/solver.rs` (a child module sees the private
// functions of its parent), it runs the reference's own code on the committed tensile mesh and prints every quantity
// the oracle's fixture `tests/golden/tensile.npz` holds, as one JSON document.
//
// How to run (in a checkout of kyle-tennison/Magnetite @ 2024_08_07):
//   1. copy this file to            src/dump_reference.rs
//   2. copy the two inputs to       tests/golden/tensile_nodes.csv, tests/golden/tensile_elements.csv
//      (from this repository's tests/golden/; written by tests/golden/make_reference_dump_inputs.py)
//   3. append to src/solver.rs:     #[cfg(test)] #[path = "dump_reference.rs"] mod dump_reference;
//   4. cargo test --release dump_reference -- --nocapture > reference_dump.txt
//   5. in this repository:          python scripts/compare_reference_dump.py reference_dump.txt
//
// What it settles besides u / f / stress / iteration count:
//   * argmin 0.10's ConjugateGradient `cost`: |r| or |r|^2 -- the dump prints state.get_cost() after ONE iteration next to
//     |b - A x1| and its square computed with nalgebra from the returned x1 (mag_options.stop_mode MAG_STOP_RNORM vs _SQ);
//   * whether `init` already reports a cost (cost of the state before the first iteration);
//   * the observer's iteration count (state.get_iter()) and best_cost of the full run (solver.rs:149-176).
use super::*;
use crate::datatypes::Vertex;

fn opt(field: &str) -> Option<f64> {
    let f = field.trim();
    if f.is_empty() {
        None
    } else {
        Some(f.parse::<f64>().expect("number"))
    }
}

fn load_nodes(path: &str) -> Vec<Node> {
    let text = std::fs::read_to_string(path).expect("tensile_nodes.csv");
    text.lines()
        .skip(1) // header: x,y,ux,uy,fx,fy  (empty field = None)
        .filter(|l| !l.trim().is_empty())
        .map(|l| {
            let c: Vec<&str> = l.split(',').collect();
            Node {
                vertex: Vertex {
                    x: c[0].trim().parse().unwrap(),
                    y: c[1].trim().parse().unwrap(),
                },
                ux: opt(c[2]),
                uy: opt(c[3]),
                fx: opt(c[4]),
                fy: opt(c[5]),
            }
        })
        .collect()
}

fn load_elements(path: &str) -> Vec<Element> {
    let text = std::fs::read_to_string(path).expect("tensile_elements.csv");
    text.lines()
        .skip(1) // header: n0,n1,n2
        .filter(|l| !l.trim().is_empty())
        .map(|l| {
            let c: Vec<usize> = l.split(',').map(|f| f.trim().parse().unwrap()).collect();
            Element {
                nodes: [c[0], c[1], c[2]],
                stress: None,
            }
        })
        .collect()
}

fn join(v: &[f64]) -> String {
    // {:e} of an f64 round-trips: the shortest digits that parse back to the same bits
    v.iter().map(|x| format!("{:e}", x)).collect::<Vec<_>>().join(",")
}

#[test]
fn dump_reference() {
    let meta = ModelMetadata {
        youngs_modulus: 69e9, // examples/tensile-example/input.json
        poisson_ratio: 0.33,
        part_thickness: 0.5,
        characteristic_length_min: 0.3,
        characteristic_length_max: 0.3,
    };
    let mut nodes = load_nodes("tests/golden/tensile_nodes.csv");
    let mut elements = load_elements("tests/golden/tensile_elements.csv");

    // ---- (1) the pieces, through the reference's private functions: K, the partition, b, and the CG state
    let mut kes = Vec::new();
    for e in elements.iter() {
        kes.push(compute_element_stiffness_matrix(
            e,
            &nodes,
            meta.poisson_ratio,
            meta.youngs_modulus,
            meta.part_thickness,
        ));
    }
    let ke0: Vec<f64> = (0..36).map(|k| kes[0][(k / 6, k % 6)]).collect();
    let k_total = build_total_stiffness_matrix(&nodes, &elements, kes);
    let (nodal_forces, nodal_displacements) = build_col_vecs(&nodes);
    let (known, unknown) = build_known_unknown_matrices(&nodal_forces, &nodal_displacements, &k_total);
    let mut b: DVector<f64> = known.column_sum();
    let known_forces: Vec<&Option<f64>> = nodal_forces.iter().filter(|x| x.is_some()).collect();
    for (i, k) in b.iter_mut().enumerate() {
        *k += known_forces[i].unwrap();
    }
    let n = unknown.nrows();
    // the same exact-zero drop and row-major push order as solver.rs:126-137 (the CSR's column order follows from it)
    let mut coo: CooMatrix<f64> = CooMatrix::new(n, n);
    for (r, c) in (0..n).flat_map(|r| (0..n).map(move |c| (r, c))) {
        let entry = unknown[(r, c)];
        if entry != 0.0 {
            coo.push(r, c, entry);
        }
    }
    let csr: CsrMatrix<f64> = CsrMatrix::from(&coo);
    let b_flat: Vec<f64> = b.iter().map(|f| *f).collect();

    let run_cg = |iters: u64| {
        let solver: ConjugateGradient<_, f64> = ConjugateGradient::new(b_flat.clone());
        let operator = ConjugateGradientOperator { a: &csr };
        Executor::new(operator, solver)
            .configure(|state| {
                state
                    .param(vec![0.0; n])
                    .max_iters(iters)
                    .target_cost(TARGET_CG_COST)
            })
            .run()
            .expect("argmin")
    };
    // the state before any iteration (does `init` report a cost?), after one iteration, and at the end
    let r0 = run_cg(0);
    let r1 = run_cg(1);
    let x1 = DVector::from_vec(r1.state().get_param().expect("x1").clone());
    let res1 = &b - &csr * &x1;
    let full = run_cg(MAX_CG_ITER);
    let best = full.state().get_best_param().expect("best_param").clone();

    // ---- (2) the entry point itself, as main.rs calls it
    run(&mut nodes, &mut elements, &meta).expect("solver::run");
    let mut u = Vec::new();
    let mut f = Vec::new();
    for nd in nodes.iter() {
        u.push(nd.ux.unwrap());
        u.push(nd.uy.unwrap());
        f.push(nd.fx.unwrap());
        f.push(nd.fy.unwrap());
    }
    let stress: Vec<f64> = elements.iter().map(|e| e.stress.unwrap()).collect();

    println!("REFERENCE_DUMP_BEGIN");
    println!("{{");
    println!("\"num_nodes\": {}, \"num_elements\": {}, \"n_free\": {}, \"nnz_ff\": {},", nodes.len(), elements.len(), n, csr.nnz());
    println!("\"ke0\": [{}],", join(&ke0));
    println!("\"b\": [{}],", join(&b_flat));
    println!("\"cost_before_first_iteration\": {:e},", r0.state().get_cost());
    println!("\"cost_after_1_iteration\": {:e},", r1.state().get_cost());
    println!("\"residual_norm_after_1_iteration\": {:e},", res1.norm());
    println!("\"residual_norm_squared_after_1_iteration\": {:e},", res1.norm_squared());
    println!("\"iterations\": {}, \"final_cost\": {:e}, \"best_cost\": {:e},", full.state().get_iter(), full.state().get_cost(), full.state().get_best_cost());
    println!("\"x_best\": [{}],", join(&best));
    println!("\"u\": [{}],", join(&u));
    println!("\"f\": [{}],", join(&f));
    println!("\"stress\": [{}]", join(&stress));
    println!("}}");
    println!("REFERENCE_DUMP_END");
}
