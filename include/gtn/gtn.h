// gtn/gtn.h -- everything a gtn caller includes, served by libgtn_amd.so.
//
// The headers below are header-only: each gtn:: entity is a thin value type or an inline
// function over the C ABI of include/gtn_amd.h (one extern "C" call per reference member
// or function; batched "_n" calls behind gtn::batched).  None of them needs hipcc -- host
// code builds with any C++17 compiler and links -lgtn_amd.
//
//   graph.h      Graph: handle with the reference's aliasing rules, node / arc accessors,
//                weights, gradient members, epsilon
//   creations.h  scalarGraph, linearGraph (implicit emissions chain), linearGraphs (batch on
//                a device tensor)
//   functions.h  compose / intersect, forwardScore / viterbiScore / viterbiPath, scalar
//                arithmetic, rational operations; gtn::batched vector forms
//   autograd.h   backward
//   utils.h      equal / isomorphic, text and binary graph files, operator<<, draw
//   rand.h       sample, randEquivalent
//   parallel.h   parallelMap for host-side work (target-graph construction)
//   batch.h      Batch: B graphs as one record, the same functions over it (criteria)
#pragma once

#include "gtn/graph.h"

#include "gtn/creations.h"
#include "gtn/functions.h"

#include "gtn/autograd.h"

#include "gtn/utils.h"
#include "gtn/rand.h"

#include "gtn/parallel.h"

#include "gtn/batch.h"
