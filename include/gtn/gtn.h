// gtn/gtn.h -- umbrella header (reference gtn/gtn.h:10-16)
#pragma once
#include "gtn/autograd.h"
#include "gtn/creations.h"
#include "gtn/functions.h"
#include "gtn/graph.h"
#include "gtn/parallel.h"
#include "gtn/rand.h"
#include "gtn/utils.h"
