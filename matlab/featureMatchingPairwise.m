function matches = featureMatchingPairwise(input, allDescriptors, numImg)
    %FEATUREMATCHINGPAIRWISE Shadows PP/featureMatching/featureMatchingPairwise.m (n x n cell, upper triangle, M x 2 double).
    %   input.useMATLABFeatureMatch = 0, input.Matchingmethod = 'Exhaustive' (the all-pairs path of the north star): all
    %   upper-triangular pairs in ONE batched device call (exhaustive 2-NN + ratio + threshold + unique).
    %   input.Matchingmethod = 'Approximate': pair by pair through the shadowed matchFeaturesScratch, as the reference's
    %   parfor does (:48-63, getMatches :103-120) - 'pca2nn' runs nearest2ApproxFloatFast on the device, 'kdtree' and
    %   'subsetpdist2' the exact device search.
    %   input.useMATLABFeatureMatch = 1 (the toolbox's matchFeatures) and binary descriptors are forwarded to the
    %   reference's own file.
    arguments
        input struct
        allDescriptors cell
        numImg (1, 1) {mustBeNumeric, mustBeFinite, mustBePositive}
    end
    isFloat = all(cellfun(@(d) isfloat(d) || isinteger(d) && ~isa(d, 'uint8'), allDescriptors(1:numImg)));
    if (isfield(input, 'useMATLABFeatureMatch') && input.useMATLABFeatureMatch == 1) || ~isFloat
        matches = aps_call_shadowed('featureMatchingPairwise', mfilename('fullpath'), input, allDescriptors, numImg);
        return;
    end
    if isfield(input, 'Matchingmethod') && strcmpi(input.Matchingmethod, 'Approximate')
        approx = 'pca2nn';
        if isfield(input, 'ApproxFloatNNMethod'), approx = input.ApproxFloatNNMethod; end
        matches = cell(numImg);
        for jj = 2:numImg
            for ii = 1:jj - 1
                m = matchFeaturesScratch(allDescriptors{ii}, allDescriptors{jj}, 'Method', 'Approximate', ...
                    'ApproxFloatNNMethod', approx, 'MatchThreshold', input.Matchingthreshold, ...
                    'MaxRatio', input.Ratiothreshold, 'Unique', true);
                matches{ii, jj} = double(m);
            end
        end
        return;
    end
    opts = struct('MaxRatio', input.Ratiothreshold, 'MatchThreshold', input.Matchingthreshold, 'Unique', 1);
    matches = aps_mex('match_pairwise', cellfun(@single, allDescriptors(1:numImg), 'UniformOutput', false), opts);
end
